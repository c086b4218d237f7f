// dmi_job.hpp — internal types shared by the translation units behind the C ABI (include/draco_mi.h):
//   dmi_job.cpp      job creation from caller-supplied tables, the encode phases of one job, the batch drivers
//   dmi_prepare.cpp  whole-mesh entry points: host + device connectivity stage, dmi_mesh_prepare / dmi_meshes_prepare / dmi_encode_mesh
// Device memory of a job (pooled chunks), pinned staging pools, the per-attribute and per-table device state.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <functional>
#include <thread>

#include "dmi_device.hpp"
#include "dmi_host.hpp"
#include "host_chains.hpp"

namespace dmi {
// Waits of the batch path that last milliseconds (a stage's device tables, a build's kernels and read-backs, an encode's chain launch): hipStreamSynchronize and
// hipEventSynchronize on a default event SPIN on a host core for as long as they wait — a quarter of a transcode's CPU time was the runtime's wait loop (sampling
// profile, scripts/experiments/transcode_sigprof.py) with six coordinator threads waiting beside sixteen walkers on a 16-CPU quota.  These POLL instead: hipEventQuery,
// a few times back to back, then with 50 µs sleeps between (events created with hipEventBlockingSync still spun in the runtime this image's torch wheel bundles: 7 % of the
// call's CPU samples stayed inside long_wait_stream).  DMI_DBG_SPIN_WAITS restores the runtime's own waits (A/B).  The short waits of a single whole-mesh call keep
// spinning: their latency is the call's.
unsigned long_wait_flags();                    // flags for hipEventCreateWithFlags (timing off)
hipError_t long_wait_event(hipEvent_t e);      // hipEventSynchronize without a spinning core
hipError_t long_wait_stream(hipStream_t s);    // returns when everything queued on s so far has finished
extern thread_local std::string g_last_error;
extern thread_local dmi_timings g_last_call;   // dmi_last_call_timings
extern thread_local size_t g_out_prefix;        // bytes the splice of the next encode on this thread leaves free in front of the attribute section (a one-shot call puts header + connectivity there: one output buffer, one copy of every stream)
extern thread_local bool g_one_shot_call;      // set by the create → encode → destroy entry points (dmi_encode_mesh[_device], dmi_encode_attributes): layout work that only pays over many encodes is skipped
inline int fail(int code, const std::string& msg) { return host_fail(code, msg); }

#define HIP_TRY(expr)                                                                                              \
  do {                                                                                                             \
    hipError_t e_ = (expr);                                                                                        \
    if (e_ != hipSuccess) {                                                                                        \
      const bool nodev = (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorInsufficientDriver || e_ == hipErrorNotInitialized); \
      return fail(nodev ? DMI_ERR_NO_DEVICE : (e_ == hipErrorOutOfMemory ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_HIP),    \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                              \
    }                                                                                                              \
  } while (0)

enum Scheme : uint8_t { kDelta = 0, kParallelogram = 1, kTexCoord = 5, kNormal = 6 };   // prediction_scheme/mod.rs:74-86
enum Transform : uint8_t { kDifference = 0, kWrapped = 1, kOctOrth = 3 };              // prediction_transform/mod.rs:92-101
enum Port : uint8_t { kToBits = 1, kCoordwise = 2, kOct = 3 };                          // portabilization/mod.rs:85-92
constexpr uint32_t kMaxPrepareWorkers = 128;   // host threads of one dmi_meshes_prepare call
constexpr uint32_t kPrepareStreams = 16;       // library streams their jobs are created on (per device)
constexpr uint32_t kDeviceRelabelMinFaces = 1u << 20;   // job creation relabels the connectivity inputs with kernels from this size up (dmi_relabel.hip): its temporaries
                                                      // are device allocations, whose release synchronises the device — a batch of mid-sized meshes on many threads must not take it
constexpr uint32_t kTileSortMinEntries = 1u << 18;   // sequences at least this long run their quantize gather tile-sorted (dmi_job.cpp) …
constexpr uint32_t kTileSortBigLog2 = 17;             // … in tiles of 2^14 … 2^17 entries, by the sequence length (dmi_job.cpp)
constexpr uint64_t kHostChainMinSymbols = 32768;   // a job whose longest stream is at least this long codes its streams on host cores (hybrid form)
// Device memory of one job comes from a few large chunks (DevPool) instead of one hipMalloc per buffer: job creation for a batch
// of meshes runs on many host threads, and ≈ 70 allocations + ≈ 20 memsets per job serialise on the runtime (17 ms of
// thread time per job before, most of it here).  A chunk is zeroed once when it is created, so pooled buffers start zeroed.
// Released job chunks are kept (per device, in power-of-two size classes, up to kChunkCacheBytes in total) and handed to the next job of
// that class: a transcode pipeline creates and destroys a thousand jobs per batch, and hipMalloc / hipFree serialise across the host
// threads that do it (hipFree also synchronises the device).  A reused chunk is zeroed again on the new job's stream.
struct ChunkCache {
  struct Item { int device; void* p; size_t cap; };
  std::mutex m;
  std::vector<Item> items;
  size_t bytes = 0;
  // what the cache may hold: dmi_process_options::device_cache_mb, default = half of the device's memory, at most 64 GiB (read once)
  static size_t limit_bytes() {
    static const size_t limit = [] {
      if (const size_t mb = device_cache_limit_mb()) return mb << 20;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || !total_b) return (size_t)16 << 30;
      return std::min<size_t>((size_t)64 << 30, total_b / 2);
    }();
    return limit;
  }
  static size_t size_class(size_t n) {   // powers of two up to 1 GiB, multiples of 256 MiB above (a 100M-triangle job is 33 GB)
    if (n > ((size_t)1 << 30)) return (n + (((size_t)1 << 28) - 1)) & ~(((size_t)1 << 28) - 1);
    size_t c = (size_t)1 << 20;
    while (c < n) c <<= 1;
    return c;
  }
  // the smallest cached chunk of `device` that holds `cap` bytes without being more than twice as large; its size comes back in `cap`
  void* acquire(int device, size_t& cap) {
    std::lock_guard<std::mutex> lock(m);
    size_t best = items.size();
    for (size_t k = 0; k < items.size(); ++k)
      if (items[k].device == device && items[k].cap >= cap && items[k].cap <= 2 * cap && (best == items.size() || items[k].cap < items[best].cap)) best = k;
    if (best == items.size()) return nullptr;
    void* p = items[best].p;
    cap = items[best].cap;
    bytes -= cap;
    items.erase(items.begin() + (long)best);
    return p;
  }
  void drop_all() {
    std::vector<Item> gone;
    { std::lock_guard<std::mutex> lock(m); gone.swap(items); bytes = 0; }
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (auto& it : gone) if (hipSetDevice(it.device) == hipSuccess) (void)hipFree(it.p);
    if (have_prev) (void)hipSetDevice(prev);
  }
  bool release(int device, void* p, size_t cap) {   // false: not kept (the caller frees it)
    std::lock_guard<std::mutex> lock(m);
    if (bytes + cap > limit_bytes()) return false;
    items.push_back({device, p, cap});
    bytes += cap;
    return true;
  }
};
extern ChunkCache g_chunk_cache;   // (process lifetime)

struct DevPool {
  struct Chunk { void* p; size_t cap, used; };
  std::vector<Chunk> chunks;
  size_t chunk_bytes = 0;
  hipStream_t stream = nullptr;
  int device = 0;
  bool zero = true;   // chunks are cleared when they are taken (job memory); temporaries skip it
  bool poison = false;   // test mode (DMI_POISON): an uncleared chunk is filled with 0xA5 — nothing may depend on what it held
  ~DevPool() { for (auto& c : chunks) if (c.p && !g_chunk_cache.release(device, c.p, c.cap)) (void)hipFree(c.p); }
  void* take(size_t n) {
    n = (n + 255) & ~(size_t)255;
    if (chunks.empty() || chunks.back().used + n > chunks.back().cap) {
      Chunk c{nullptr, ChunkCache::size_class(std::max(n, chunk_bytes)), 0};
      c.p = g_chunk_cache.acquire(device, c.cap);
      if (!c.p && hipMalloc(&c.p, c.cap) != hipSuccess) {   // out of memory with chunks parked in the cache: hand them back and try once more
        (void)hipGetLastError();
        g_chunk_cache.drop_all();
        if (hipMalloc(&c.p, c.cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      }
      if ((zero || poison) && hipMemsetAsync(c.p, zero ? 0 : 0xA5, c.cap, stream) != hipSuccess) { (void)hipFree(c.p); return nullptr; }
      chunks.push_back(c);
    }
    Chunk& c = chunks.back();
    void* p = static_cast<uint8_t*>(c.p) + c.used;
    c.used += n;
    return p;
  }
};
extern thread_local DevPool* g_active_pool;   // set while dmi_job_create runs on this thread

struct DevMem {
  void* p = nullptr;
  size_t bytes = 0;
  bool pooled = false;
  ~DevMem() { if (p && !pooled) (void)hipFree(p); }
  int alloc(size_t n) {
    if (p && !pooled) (void)hipFree(p);
    p = nullptr; pooled = false;
    bytes = n;
    if (n == 0) return DMI_OK;
    hip_used().store(true, std::memory_order_relaxed);
    if (g_active_pool) {
      p = g_active_pool->take(n);
      if (!p) return host_fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (job pool)");
      pooled = true;
      return DMI_OK;
    }
    HIP_TRY(hipMalloc(&p, n));
    return DMI_OK;
  }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// A view into the job's slab (same accessors as DevMem, no ownership)
struct SlabView {
  void* p = nullptr;
  size_t bytes = 0;
  template <class T> T* as() const { return static_cast<T*>(p); }
};

struct TableDev {
  uint32_t F = 0, V = 0, n_seq = 0;
  DevMem fan_hdr, fan_apex, fan;   // fan rows (only for tables with a fused sweep)
  DevMem s2p_sorted, sorted_dest;   // tile-sorted form of the quantize gather: slot j → point / sequence index (optional)
  DevMem c2r, opp, seq, s2p;   // c2r = corner → sequence index of its vertex; s2p = sequence index → point (= point_idx(seq[i]))
  DevMem frec;                  // one-shot jobs in the mesh's own face order (round 6): 32-byte face records {ranks[3], -, opposite corners[3], -} instead of c2r (launch_face_records)
  const uint32_t* s2p_host = nullptr;   // (during job creation, host-relabel form) the same array in the upload staging
  // sharing: a table whose arrays equal another table's reuses its device copies
  int alias_of = -1;
};

struct AuxInfo { uint8_t zero_prob = 0; uint32_t count = 0; int rans_desc = -1; int desc = -1; };

struct AttJob {
  dmi_attribute desc{};
  Scheme scheme = kDelta;
  Transform transform = kDifference;
  Port port = kCoordwise;
  int bits = 11;
  int nq = 0;            // components after portabilization
  int table = 0;         // index into tables
  int parent = -1;
  int fused_into = -1;   // ≥ 0: predicted by the fused seam-free sweep launched for that position attribute
  int fused_nrm = -1, fused_uv = -1;   // (on a position attribute) the attributes its fused sweep also predicts
  int qfmt = QF_I32;     // layout of qs (QFmt): packed for the attributes of a fused sweep whose widths allow it
  bool sym16 = false;    // symbols stored as uint16 (alphabet bound ≤ 65536)
  DevMem raw, s2v /*sequence index → value index, only with a point_to_value map*/, qs, sym, aux /*flips or orient*/, rtable, rec, out, partials, ipartials;
  // views into dmi_job::slab — one memset, one read-back per encode: small = 16 scratch words (minmax[2], counters[2], flags[2], …,
  // out_len[2]*2, ticks[2]), meta = quantization ranges, hist = symbol histogram (bins_cap words), summary = orientation chunk summaries
  SlabView small, meta, hist, summary;
  size_t slab_off = 0;
  uint32_t bins_cap = 0;
  uint32_t bins = 0;
  uint64_t n_sym = 0;
  uint64_t out_cap = 0, aux_cap = 0;
  DevMem aux_out, aux_rec, chunk_info, batch_flags, aux_flags;
  DevMem flip_partials;   // normals: per-block counts of the sweep (kSweepMaxBlocks words)
  DevMem fix_list;        // texture coordinates of a fused sweep: the entries it defers to k_texcoord_fixup (≤ one per sequence entry)
  uint32_t flip_blocks = 0;   // blocks of the sweep launched this encode
  DevMem aux_bits;   // host-core chains: the compacted orientation transition bits (1 byte each)
  DevMem freq, hdr, aux_entries;   // device form of the table stage: normalised-frequency scratch, serialised table, rABS record pair
  uint32_t hdr_cap = 0;
  DevMem fan_hdr, fan_apex, fan;   // fan rows of a normal attribute swept on its own table (position ranks, centre in apex)
  FreqTable ft;
  std::vector<RansEntry> rt_host;     // staging (kept alive until the copies have been issued)
  std::vector<uint32_t> info_host;
};
}  // namespace dmi
using namespace dmi;   // (internal header: every includer is a library translation unit)

// hipStreamCreate costs ≈ 1 ms and serialises across host threads (31 ms per job with 32 creator threads): jobs created by
// dmi_meshes_prepare share one library-owned stream per worker thread, kept for the life of the process.
struct StreamHolder {
  hipStream_t s = nullptr;
  ~StreamHolder() { if (s) (void)hipStreamDestroy(s); }
};
extern thread_local std::shared_ptr<StreamHolder> g_adopt_stream;   // set by a dmi_meshes_prepare worker around dmi_job_create

namespace dmi { struct EarlyQuant; struct SeqStream; }   // (below, behind TempDev)
struct dmi_job {
  dmi_config cfg{};
  std::shared_ptr<EarlyQuant> early;   // (whole-mesh one-shot calls: see EarlyQuant)
  std::shared_ptr<SeqStream> seq_stream;   // (whole-mesh one-shot calls: the sequence array shipped during the walk — tables[0].seq views its memory)
  dmi_debug debug{};                   // the switches of the call that created the job (cfg.debug points here): dmi_job_encode / dmi_jobs_encode work under them
  hipStream_t stream = nullptr;
  std::shared_ptr<StreamHolder> stream_owner;   // set when the library created the stream
  DevPool pool;   // (declared before every DevMem of the job: destroyed after them)
  DevPool donated;   // chunks taken over from the connectivity stage of a one-shot call (DeviceTableView::donor): released with the job
  std::vector<AttJob> atts;
  std::vector<TableDev> tables;
  DevMem upload_region;   // every array job creation uploads, in one piece (host-relabel form): filled through one pinned staging copy
  DevMem descs;
  DevMem slab;   // small / meta / hist / summary of every attribute, laid out exactly like the pinned read-back buffer
  void* pinned = nullptr;   // host-pinned readback area
  size_t pinned_bytes = 0;
  hipEvent_t ev[8]{};
  bool have_events = false;
  dmi_timings last{};
  float create_device_ms = 0;   // device span of job creation (hipEvents on the job's stream; DMI_FLAG_TIMINGS)
  uint32_t last_fixups = 0;   // texture-coordinate entries the last encode's fused sweep deferred to k_texcoord_fixup
  uint64_t predict_bytes = 0;
  hipGraphExec_t graph_a = nullptr;   // phase A captured once (launch-bound for small meshes)
  bool graph_tried = false;
  uint8_t* readback = nullptr;     // where the slab of the current encode was read back to (pinned, or a batch arena slot)
  bool dev_tables = false;         // tables, metadata parameters and chain descriptors are produced on the device (k_tables): no host round trip
                                   // between the histograms and the chains (DMI_HOST_TABLES=1 or a ToBits attribute keep the host form)
  uint8_t* out_pinned = nullptr;   // grow-only pinned arena for the coded bytes of one encode
  size_t out_pinned_cap = 0;
  // Hybrid form (host_chains.cpp): the streams of a single large mesh are coded on host cores from the device-built symbols and tables.
  bool host_chains = false;
  struct HostStage* stage = nullptr;                 // pinned staging of symbols / tables / metadata bits (process-wide pool)
  std::vector<std::unique_ptr<HostChainOut>> host_out;   // [2·i] rANS stream of attribute i, [2·i + 1] its metadata rABS stream
  std::vector<hipEvent_t> copy_ev;                   // "attribute i has arrived"
  struct Run {   // state carried between the phases of one encode
    std::vector<size_t> rans_off, aux_off;   // offsets into out_pinned
    std::vector<const uint8_t*> rans_ptr, aux_ptr;   // host addresses of the coded bytes (pinned memory)
    std::vector<uint32_t> rans_len, aux_len;
    std::vector<const uint8_t*> hdr_ptr;     // serialised frequency tables (host form: FreqTable::header; device form: read back)
    std::vector<uint32_t> hdr_len;
    std::vector<size_t> hdr_off;
    struct Pending { void* dst; const void* src; size_t bytes; };
    std::vector<Pending> pending;   // host → device copies deferred to the batch driver (plan mode of phase B)
    std::vector<size_t> pin_off;
    std::vector<AuxInfo> aux;
    std::vector<ChainDesc> descs;
  } run;
  ~dmi_job();
  void release() {
    if (stream) (void)long_wait_stream(stream);   // (the job's device memory goes back to a cache, not through a synchronising hipFree)
    if (pinned) (void)hipHostFree(pinned);
    if (out_pinned) (void)hipHostFree(out_pinned);
    if (graph_a) (void)hipGraphExecDestroy(graph_a);
    if (have_events) for (auto& e : ev) (void)hipEventDestroy(e);
    for (auto& e : copy_ev) (void)hipEventDestroy(e);
  }
};

// Pinned host staging for the hybrid form, pooled for the life of the process: pinning ≈ 150 MB costs tens of milliseconds, which a
// create → encode → destroy call (dmi_encode_attributes) would otherwise pay every time.
struct HostStage {
  int device = -1;
  uint8_t* p = nullptr;
  size_t cap = 0;
  bool in_use = false;
  bool registered = false;   // malloc + hipHostRegister (huge pages) rather than hipHostMalloc
};

// ---- the encode phases of one job (dmi_encode.cpp), shared with the batch drivers (dmi_batch.cpp) ----
int check_value_bounds(const dmi::AttJob& a, const uint32_t* small, uint32_t i);
int encode_phase_a(dmi_job* job, bool plan_only = false);
int encode_phase_b(dmi_job* job, bool plan_only = false, bool host_chains = false);
uint32_t count_streams(const dmi_job* job);
int encode_phase_b_dev(dmi_job* job, dmi::ChainDesc* desc_base, dmi::ChainDesc* hdr_desc_base, bool host_chains = false);
int check_device_flags(const uint32_t* small, uint32_t i);
int encode_phase_c_packed(dmi_job* job, const dmi::PackEntry* table, uint32_t first_desc, const uint8_t* arena_host);
int encode_phase_c3(dmi_job* job, dmi_buffer* out);
int run_phase_a(dmi_job* job);
HostStage* acquire_stage(int device, size_t bytes);
void release_stage(HostStage* st);
namespace dmi {
int to_buffer(const std::vector<uint8_t>& v, dmi_buffer* out);
// the universal corner table of a mesh as the device connectivity stage left it in HBM (mesh-local ids, the mesh's own numbering)
struct DeviceTableView { const uint32_t* c2p; const uint32_t* c2v; const uint32_t* opp; bool trusted_sequences;
                         bool values_on_device = false;   // dmi_attribute::values are device pointers (dmi_encode_mesh_device): copied device to device
                         // attribute corner tables the device built (k_att_*): a table whose host corner_to_vertex array is att_key[k] has its device
                         // copies at att_c2v[k] / att_opp[k] (the deferred relabelling reads them there: nothing is uploaded)
                         uint32_t n_att = 0; const uint32_t* const* att_key = nullptr; const uint32_t* const* att_c2v = nullptr; const uint32_t* const* att_opp = nullptr;                         // (one-shot calls) the pool `opp` lives in: a job that uses the array as it is (the mesh's own face order) takes the pool's chunks instead of copying 12 bytes per face
                         struct DevPool* donor = nullptr;
};
// Job creation as part of a batch (dmi_meshes_prepare): the job is planned and its memory laid out on a worker thread, but every piece of
// device work is only RECORDED here — the coordinator runs the uploads, the relabelling, the fan rows and the map compositions of all
// jobs in one launch per kernel (dmi_prepare.cpp).  Requires device-resident tables and a mesh whose tables are all the universal one.
struct JobDefer {
  // in
  hipStream_t stream = nullptr;                 // the coordinator's stream (the job's chunk is cleared on it)
  std::vector<const void*> values_dev;          // per attribute: its raw values, already in device memory (nullptr: take them from the host pointer now)
  std::vector<const uint32_t*> maps_dev;        // per attribute: its point → value map in device memory (nullptr = none)
  // out
  struct Copy { void* dst; const void* src_dev; size_t bytes; };
  std::vector<Copy> copies;                     // device → device: values into the job's buffers
  struct Clear { void* p; size_t bytes; };      // (multiples of 4 bytes)
  std::vector<Clear> clears;                    // ranges of the job's (uncleared) chunk that must start as zeros: one launch for all jobs
  // One RelabelItem per distinct corner table of the job, the universal one first.  seq = HOST pointer of the table's sequence (the
  // coordinator uploads it); an attribute table of its own (interior seams) also carries HOST pointers of its vertex ids / opposite
  // corners in host_c2v / host_opp (uploaded by the coordinator: c2v / opp of the item are then null until it has placed them);
  // order_item = index WITHIN relabels of the table whose face order the item follows (0); offsets filled by the coordinator.
  std::vector<RelabelItem> relabels;
  std::vector<const uint32_t*> host_c2v, host_opp;   // parallel to relabels (null: the table is resident)
  std::vector<FanItem> fans;                    // off filled by the coordinator
  std::vector<ComposeItem> compose;
};
int job_create_impl(const dmi_attribute* atts, const dmi_corner_table* tables, uint32_t n_atts, const uint32_t* seeds, uint32_t n_seeds,
                    const dmi_config* cfg, const DeviceTableView* dev, dmi_job** job, JobDefer* defer = nullptr);
// Device temporaries of job creation / the connectivity stage: bump-allocated from cached chunks like a job's own memory (hipMalloc and
// hipFree serialise across host threads, and hipFree waits for the device), not cleared; handed back when the work on `stream` is over.
struct TempDev {
  DevPool pool;
  void init(int device, hipStream_t s, size_t bytes_hint) { pool.device = device; pool.stream = s; pool.chunk_bytes = bytes_hint; pool.zero = false; }
  bool owner_waits = false;   // the owner itself makes sure the work on the chunks is over before this is destroyed (an event it waits for): no synchronisation here
  ~TempDev() { if (!owner_waits && !pool.chunks.empty()) (void)long_wait_stream(pool.stream); }   // (nothing may still be using a chunk when the cache hands it to the next taker)
  template <class T> T* take(size_t n) { return static_cast<T*>(pool.take((n ? n : 1) * sizeof(T))); }
};

// Early stage of a whole-mesh call whose values are already in HBM (dmi_encode_mesh_device): value ranges and the quantization in VALUE order, into the
// packed layouts of a fused sweep, issued on a side stream BEFORE the host's serial walks — the device has ≈ 90 ms of nothing to do there — so that
// the pass, once a sequence exists, only gathers one 16-byte record per entry into coding order (dmi_kernels.hip k_seq_gather_rec).  Meshes whose
// attributes are all per-point (no point → value map).  Adopted by the job the call
// creates when the job's plan (formats, bits) is the one guessed here; dropped otherwise (the job then quantizes as always).
struct EarlyQuant {
  int device = 0;
  hipStream_t stream = nullptr;
  TempDev mem;
  hipEvent_t t0 = nullptr, t1 = nullptr;   // around the early kernels (t1 = "the packed values are there")
  struct Att { const void* values; uint32_t n; int N, kind, fmt, bits; uint8_t* slot /* [small 64 B][meta 64 B] like a job's slab slot */; };
  void* rec = nullptr;   // n records of 16 bytes: the quantized position | texture coordinate | normal of value v (dmi_kernels.hip QuantRec)
  std::vector<Att> atts;
  int att_of_kind[3] = {-1, -1, -1};              // attribute index of the position / normal / texture coordinate
  int32_t* ipartials[3] = {nullptr, nullptr, nullptr};   // per-block joint i32 min/max pairs of k_value_quantize_rec, by kind; folded by the consumer's first block
  uint32_t ipartial_blocks = 0;
  const uint32_t* nrm_flags = nullptr;   // per block of the quantizer: a zero-length normal seen (ipartial_blocks words)
  // EarlySlots of a consumer kernel: the stage's slots → the slab slots of `job_atts` (dst_of(attribute index) = its `small` words)
  template <class F> EarlySlots slots_for(F dst_of) const {
    EarlySlots es{};
    for (int k = 0; k < 3; ++k) {
      if (att_of_kind[k] < 0) continue;
      es.src[k] = reinterpret_cast<const uint32_t*>(atts[(size_t)att_of_kind[k]].slot);
      es.dst[k] = dst_of((size_t)att_of_kind[k]);
      es.ipartials[k] = ipartials[k];
    }
    es.ipartial_blocks = ipartial_blocks;
    es.nrm_flags = att_of_kind[1] >= 0 ? nrm_flags : nullptr;
    return es;
  }
  ~EarlyQuant() {
    if (stream) (void)long_wait_stream(stream);
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
  }
};
extern thread_local std::unique_ptr<EarlyQuant> g_early_quant;   // set by dmi_encode_mesh_device around mesh_prepare_impl; taken by job_create_impl

// The universal sequence of a whole-mesh call shipped to the device WHILE the sequencer writes it (round 6).  Job creation used to start with the upload of
// the finished sequence — 20 MB, ≈ 0.4 ms of a 0.9 ms device span per 10M triangles, with the device idle through the 35 ms the sequencer took to write it.
// The walk publishes its progress every 2^16 entries (SeqProgress); a thread of this object copies what is new to a device array on a stream of its own;
// job creation takes the array over (it only waits for the last piece).  Set by dmi_encode_mesh_device for large one-shot calls, used by
// build_connectivity (the universal sequencer), adopted by job_create_impl when the sequence it is handed IS the streamed one.
struct SeqStream {
  int device = 0;
  hipStream_t stream = nullptr;
  TempDev mem;
  uint32_t* d_seq = nullptr;
  uint32_t cap = 0;                 // entries d_seq holds
  SeqProgress progress;
  std::atomic<uint32_t> final_n{0xFFFFFFFFu};   // set by the sequencer's caller when the walk is over
  std::atomic<bool> failed{false};
  dmi::Thread uploader;
  std::mutex wake_mutex; std::condition_variable wake;   // (the walk's end wakes the uploader at once)
  hipEvent_t ev = nullptr;          // recorded behind the last piece
  const uint32_t* host = nullptr;   // (after join) the array that was shipped
  uint32_t n = 0;                   // … and its length
  bool start(int dev, hipStream_t s, uint32_t capacity);
  void finish(uint32_t n_entries);  // the walk is over: ship the rest, record `ev`, join
  ~SeqStream();
};
extern thread_local std::shared_ptr<SeqStream> g_seq_stream;
int early_quantize_issue(const dmi_attribute* atts_dev, uint32_t n_atts, const dmi_config& cfg, hipStream_t side, std::unique_ptr<EarlyQuant>& out);


// Attribute corner tables of one connectivity group on the device (dmi_conn.hip k_att_*): one item per (mesh of the group, non-position
// attribute) whose point → value map is not the position map entry for entry (such an attribute has no seam but the boundary).  The tables
// stay on the device for the batched relabelling and come back for the host's walks (seam flags → the Edgebreaker's seam streams, the
// table → the attribute's sequencer).  Methods: dmi_prepare.cpp.
struct AttStage {
  struct Item { uint32_t member /* the mesh's ConnMeshDesc */, k /* index among its non-position attributes */, corner_off, vert_off, F; };
  std::vector<Item> items;            // sorted by member
  std::vector<AttItemDesc> descs;
  size_t corners = 0, verts = 0;
  uint32_t *d_c2v = nullptr, *d_opp = nullptr, *d_lmc = nullptr;
  size_t rb_items = 0, rb_info = 0, rb_seam = 0, rb_c2v = 0, rb_opp = 0, rb_lmc = 0;   // offsets in the group's host staging
  const uint8_t* hp = nullptr;        // that staging (set by issue)
  // Round 6: what comes back.  The attribute's opposite corners are NOT read back (opposite_att[c] = seam[c] ? none : opposite[c]: the host forms them from
  // the universal table and the seam flags it has anyway — 12 bytes per face less on the link) unless want_opp (dmi_device_attribute_table hands them
  // to its caller); of the left-most corners — one per ATTRIBUTE vertex, all items' back to back, their number only known once the kernels have run — the
  // first lmc_first entries come back with the rest of the stage (2 × the universal vertices + 4096: an attribute has its universal vertices plus one per
  // seam crossing) and complete() fetches what lies beyond once the counts are in: 2–3 bytes per face instead of the worst case's 12.
  bool want_opp = false;
  size_t lmc_first = 0;
  hipStream_t stream = nullptr;
  int complete();                     // after the stage's event: the left-most corners beyond lmc_first, if any (a blocking copy; rare)
  void add(uint32_t member, uint32_t k, uint32_t F, uint32_t vcap, uint32_t map_off_words);
  size_t layout(size_t at);           // places the read-back regions from offset `at` on; returns their end
  size_t device_bytes() const;
  int issue(const ConnArgs& a, TempDev& mem, uint8_t* host, hipStream_t s);   // after launch_conn_tables on s
};

// ---- device-built meshes (dmi_build.cpp → dmi_prepare.cpp) ----
// The meshes one group of dmi_meshes_build produced, resident on `device`: arena A = the faces of all members as ONE array (member after
// member: the layout the batched connectivity kernels index) followed by the point → value maps, arena B = the unique values.  A host copy
// of arena A (the serial walks read faces and maps) — and of arena B on request — lives in pinned staging.  Shared by its members' owners.
struct BuiltGroup {
  int device = 0;
  hipStream_t stream = nullptr;
  TempDev keep;
  uint8_t* d_base = nullptr;          // arena A at offset 0, arena B at b_off
  size_t a_bytes = 0, b_off = 0, b_bytes = 0;
  HostStage* stage = nullptr;
  uint8_t *h_a = nullptr, *h_b = nullptr;
  struct Att { size_t val_off = 0, map_off = (size_t)-1; uint32_t n_unique = 0; uint8_t att_type = 0; };   // byte offsets from d_base (maps also from h_a)
  struct Member { uint32_t F = 0, P = 0, raw_faces = 0; size_t faces_off = 0 /* bytes into arena A */; std::vector<Att> atts /* in the built mesh's order */; };
  std::vector<Member> members;        // in arena order
  uint64_t total_faces = 0;           // Σ members' F
  // The universal corner tables of all members (dmi_conn.hip), issued by dmi_meshes_build right behind the build so that they are on the
  // host by the time dmi_built_meshes_prepare's walks want them (a transcode pipeline prepares stage k while stage k+1 is built).
  struct Conn {
    bool issued = false, any_mapped = false;
    bool quad = false;   // the read-back `opp` holds 4·face + k ids (no member has a point → value map: the walks' quad class, CornerTables::quad)
    TempDev mem;
    HostStage* stage = nullptr;
    uint8_t* hp = nullptr;
    uint32_t *d_c2v = nullptr, *d_opp = nullptr;
    size_t rb_opp = 0, rb_c2v = 0, rb_lmc = 0, rb_onb = 0, rb_words = 0;
    uint32_t n_desc = 0;
    uint64_t total_verts = 0;
    hipEvent_t ev = nullptr;
    AttStage att;   // attribute corner tables of the members' attributes (k_att_*)
  } conn;
  BuiltGroup() = default;
  BuiltGroup(const BuiltGroup&) = delete;
  BuiltGroup& operator=(const BuiltGroup&) = delete;
  ~BuiltGroup() {
    if (conn.ev) { (void)hipEventSynchronize(conn.ev); (void)hipEventDestroy(conn.ev); }
    if (stream) (void)long_wait_stream(stream);
    release_stage(conn.stage);
    release_stage(stage);
  }
};
// the connectivity kernels + read-back of a built group on `s` (dmi_prepare.cpp); bg.conn.ev fires when the tables are on the host
int built_group_issue_tables(BuiltGroup& bg, hipStream_t s);
struct BuiltDevice : BuiltBase {
  std::shared_ptr<BuiltGroup> group;
  uint32_t member = 0;
  std::vector<dmi_attribute> views;        // host view (values null unless they were read back)
  std::vector<std::vector<uint32_t>> parents;
};
// library streams / NUMA placement shared by the whole-mesh translation units (defined in dmi_prepare.cpp)
std::shared_ptr<StreamHolder> library_thread_stream(int device);
hipStream_t library_group_stream(int device, int which);
struct NumaScope { void* impl = nullptr; explicit NumaScope(int device); ~NumaScope(); NumaScope(const NumaScope&) = delete; NumaScope& operator=(const NumaScope&) = delete; };
}  // namespace dmi
