// dmi_job.cpp — C ABI basics of libdraco_mi.so (include/draco_mi.h: errors, buffers, cached memory) and JOB CREATION: a mesh's attributes
// and corner tables become a resident job — inputs uploaded once, connectivity relabelled into coding order, fan rows, every buffer of
// the encode pipeline laid out in one pooled chunk.  The encode itself is dmi_encode.cpp, batches dmi_batch.cpp, whole-mesh calls
// dmi_prepare.cpp.  There is NO CPU fallback: without a HIP device every encode entry point returns DMI_ERR_NO_DEVICE.
#include "dmi_job.hpp"

namespace dmi {
thread_local std::string g_last_error;
thread_local dmi_timings g_last_call{};
thread_local bool g_one_shot_call = false;
thread_local size_t g_out_prefix = 0;
int host_fail(int code, const std::string& msg) { g_last_error = msg; return code; }   // shared with the host-only translation units
ChunkCache g_chunk_cache;
thread_local DevPool* g_active_pool = nullptr;
}  // namespace dmi

using namespace dmi;

thread_local std::shared_ptr<StreamHolder> g_adopt_stream;
namespace dmi { thread_local std::unique_ptr<EarlyQuant> g_early_quant; thread_local std::shared_ptr<SeqStream> g_seq_stream; }

bool dmi::SeqStream::start(int dev, hipStream_t s, uint32_t capacity) {
  device = dev; stream = s; cap = capacity;
  if (!s || !capacity) return false;
  mem.init(dev, s, (size_t)capacity * 4 + 4096);
  d_seq = mem.take<uint32_t>(capacity);
  if (!d_seq) return false;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; return false; }
  uploader = dmi::Thread([this] {
    if (hipSetDevice(device) != hipSuccess) { failed.store(true); return; }
    uint32_t sent = 0;
    constexpr uint32_t kPiece = 1u << 20;   // 4 MiB pieces; the thread wakes once a millisecond (or when the walk is over): a few dozen wake-ups per 10M-triangle mesh
    for (;;) {
      const uint32_t fin = final_n.load(std::memory_order_acquire);
      const uint32_t have = fin != 0xFFFFFFFFu ? fin : progress.written.load(std::memory_order_acquire);
      const uint32_t* h = progress.host.load(std::memory_order_acquire);
      if (h && have > cap) { failed.store(true); return; }
      if (h && (have - sent >= kPiece || (fin != 0xFFFFFFFFu && have > sent))) {
        if (hipMemcpyAsync(d_seq + sent, h + sent, (size_t)(have - sent) * 4, hipMemcpyHostToDevice, stream) != hipSuccess) { (void)hipGetLastError(); failed.store(true); return; }
        sent = have;
        continue;
      }
      if (fin != 0xFFFFFFFFu) break;
      std::unique_lock<std::mutex> lock(wake_mutex);
      wake.wait_for(lock, std::chrono::milliseconds(1), [&] { return final_n.load(std::memory_order_acquire) != 0xFFFFFFFFu; });
    }
    host = progress.host.load(); n = sent;
    if (hipEventRecord(ev, stream) != hipSuccess) { (void)hipGetLastError(); failed.store(true); }
  });
  return true;
}
void dmi::SeqStream::finish(uint32_t n_entries) {
  { std::lock_guard<std::mutex> lock(wake_mutex); final_n.store(n_entries, std::memory_order_release); }
  wake.notify_all();
  if (uploader.joinable()) uploader.join();
}
dmi::SeqStream::~SeqStream() {
  if (uploader.joinable()) { { std::lock_guard<std::mutex> lock(wake_mutex); final_n.store(0, std::memory_order_release); } wake.notify_all(); uploader.join(); }
  if (stream && d_seq) (void)hipStreamSynchronize(stream);   // (nothing may still be copying into the array when its chunk goes back to the cache)
  if (ev) (void)hipEventDestroy(ev);
}

static std::mutex g_stage_mutex;
static std::vector<HostStage*> g_stages;   // (never freed: process-lifetime staging)
// Staging memory is ordinary host memory on transparent huge pages, registered with the runtime (page-locked, DMA at pinned-memory speed):
// the host's serial walks read the device-built tables straight out of it, and with 4 KiB pages — what hipHostMalloc hands out — nearly
// every step of such a walk misses the TLB (Edgebreaker traversal of the 10M-triangle grid: 78 ms on huge pages, 110–130 ms without).
// hipHostMalloc remains the fallback where registration fails.
static void stage_free(HostStage* st) {
  if (st->p) { if (st->registered) { (void)hipHostUnregister(st->p); std::free(st->p); } else (void)hipHostFree(st->p); }
  st->p = nullptr; st->cap = 0; st->registered = false;
}
static bool stage_alloc(HostStage* st, size_t want) {
  const bool plain = (process_flags() & DMI_PROCESS_NO_THP) != 0;
  void* q = nullptr;
  if (!plain && want >= ((size_t)4 << 20) && posix_memalign(&q, (size_t)2 << 20, want) == 0 && q) {
    advise_huge_pages(q, want);
    if (hipHostRegister(q, want, hipHostRegisterDefault) == hipSuccess) { st->p = static_cast<uint8_t*>(q); st->cap = want; st->registered = true; return true; }
    (void)hipGetLastError();
    std::free(q);
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&st->p), want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); st->p = nullptr; return false; }
  st->cap = want; st->registered = false;
  return true;
}
HostStage* acquire_stage(int device, size_t bytes) {
  hip_used().store(true, std::memory_order_relaxed);
  HostStage* best = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    for (HostStage* st : g_stages) {   // the smallest free stage that fits, else the largest free one (it is grown below)
      if (st->in_use || st->device != device) continue;
      const bool fits = st->cap >= bytes, best_fits = best && best->cap >= bytes;
      if (!best || (fits && (!best_fits || st->cap < best->cap)) || (!fits && !best_fits && st->cap > best->cap)) best = st;
    }
    if (!best) { best = new HostStage(); best->device = device; g_stages.push_back(best); }
    best->in_use = true;
  }
  if (best->cap < bytes) {
    stage_free(best);
    const size_t want = ChunkCache::size_class(bytes);   // (size classes: a worker's next mesh rarely makes its stage grow again)
    if (!stage_alloc(best, want)) {
      {   // pinned memory exhausted with idle stages parked: release them and try once more
        std::lock_guard<std::mutex> lock(g_stage_mutex);
        for (HostStage* st : g_stages) if (!st->in_use && st->p) stage_free(st);
      }
      if (!stage_alloc(best, want)) { std::lock_guard<std::mutex> lock(g_stage_mutex); best->in_use = false; return nullptr; }
    }
  }
  return best;
}
void release_stage(HostStage* st) {
  if (!st) return;
  std::lock_guard<std::mutex> lock(g_stage_mutex);
  st->in_use = false;
}
dmi_job::~dmi_job() {
  release();
  release_stage(stage);
}
namespace {

int upload(DevMem& m, const void* src, size_t bytes, hipStream_t s, hipMemcpyKind kind = hipMemcpyHostToDevice) {
  int rc = m.alloc(bytes);
  if (rc) return rc;
  if (bytes) HIP_TRY(hipMemcpyAsync(m.p, src, bytes, kind, s));
  return DMI_OK;
}

int validate_and_plan(const dmi_attribute* atts, uint32_t n_atts, const dmi_config& cfg, std::vector<AttJob>& out) {
  out.resize(n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = out[i];
    a.desc = atts[i];
    const dmi_attribute& d = atts[i];
    if (d.num_components < 1 || d.num_components > 4) return fail(DMI_ERR_UNSUPPORTED_NUM_COMPONENTS, "attribute " + std::to_string(i) + ": components must be 1..4");
    // GroupConfig::default_for, attribute_encoder.rs:59-108
    switch (d.att_type) {
      case DMI_ATT_POSITION: a.scheme = kParallelogram; a.transform = kWrapped; break;
      case DMI_ATT_NORMAL: a.scheme = kNormal; a.transform = kOctOrth; break;
      case DMI_ATT_TEXCOORD: a.scheme = kTexCoord; a.transform = kWrapped; break;
      case DMI_ATT_CUSTOM: a.scheme = kParallelogram; a.transform = kWrapped; break;
      default: a.scheme = kDelta; a.transform = kDifference; break;
    }
    if (d.att_type == DMI_ATT_POSITION && cfg.pos_scheme == 0xD0) { a.scheme = kDelta; a.transform = kDifference; }
    // portabilization::Config::default_for, portabilization/mod.rs:126-142
    a.port = d.att_type == DMI_ATT_NORMAL ? kOct : (d.att_type == DMI_ATT_CUSTOM ? kToBits : kCoordwise);
    a.bits = d.att_type == DMI_ATT_POSITION ? (cfg.pos_bits ? cfg.pos_bits : 11)
             : d.att_type == DMI_ATT_TEXCOORD ? (cfg.uv_bits ? cfg.uv_bits : 10)
             : d.att_type == DMI_ATT_NORMAL ? 8 : (cfg.generic_bits ? cfg.generic_bits : 11);
    if (a.bits < 1 || a.bits > 30) return fail(DMI_ERR_INVALID_ARGUMENT, "quantization bits out of range");
    if (a.port == kToBits) {
      if (d.component_type != DMI_U32 && d.component_type != DMI_I32 && d.component_type != DMI_F32)
        return fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "ToBits needs a 4-byte component type");
      a.nq = d.num_components;
    } else {
      if (d.component_type != DMI_F32) return fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "only f32 attributes are quantized on the device");
      a.nq = (a.port == kOct) ? 2 : d.num_components;
      if (a.port == kOct && d.num_components != 3) return fail(DMI_ERR_UNSUPPORTED_NUM_COMPONENTS, "normals need 3 components");
    }
    a.parent = -1;
    if (a.scheme == kNormal || a.scheme == kTexCoord) {
      // parents are looked up among already-encoded attributes (encode/attribute/mod.rs:63-66)
      if (d.parent_index < 0) return fail(DMI_ERR_BAD_PARENT, "attribute " + std::to_string(i) + " needs a Position parent");
      if ((uint32_t)d.parent_index >= i) return fail(DMI_ERR_PARENT_NOT_ENCODED, "parent attribute is not encoded before its child");
      const dmi_attribute& p = atts[d.parent_index];
      if (a.scheme == kNormal && p.att_type != DMI_ATT_POSITION) return fail(DMI_ERR_BAD_PARENT, "normal prediction needs a Position parent");
      if (out[d.parent_index].nq != 3) return fail(DMI_ERR_BAD_PARENT, "parent attribute must have 3 components");
      if (a.nq != 2) return fail(DMI_ERR_UNSUPPORTED_NUM_COMPONENTS, "texture coordinates / normals must portabilize to 2 components");
      a.parent = d.parent_index;
    } else if (d.parent_index >= 0 && (uint32_t)d.parent_index >= i) {
      return fail(DMI_ERR_PARENT_NOT_ENCODED, "parent attribute is not encoded before its child");
    }
    a.table = (int)i;
  }
  return DMI_OK;
}

}  // namespace


namespace {

uint32_t symbol_bins(const AttJob& a) {
  // upper bound of (largest symbol + 1) from the quantizer's range:
  //   wrapped difference: |corr| ≤ max_diff/2 ≤ 2^(bits-1) → zig-zag ≤ 2^bits            → bins 2^bits + 2
  //   plain difference:   |corr| ≤ 2^bits - 1             → zig-zag ≤ 2^(bits+1) - 1     → bins 2^(bits+1)
  //   oct-orthogonal:     corr ∈ [0, 255]                                                 → bins 256 (+slack)
  if (a.transform == kOctOrth) return 512;
  if (a.port == kToBits) return 0;   // decided after the min/max readback
  if (a.transform == kWrapped) return (1u << a.bits) + 2;
  return 1u << (a.bits + 1);
}

}  // namespace

extern "C" {

const char* dmi_strerror(int s) {
  switch (s) {
    case DMI_OK: return "ok";
    case DMI_ERR_INVALID_ARGUMENT: return "invalid argument";
    case DMI_ERR_UNSUPPORTED_DATA_TYPE: return "unsupported data type";
    case DMI_ERR_UNSUPPORTED_NUM_COMPONENTS: return "unsupported number of components";
    case DMI_ERR_PARENT_NOT_ENCODED: return "parent attribute not encoded yet";
    case DMI_ERR_BAD_PARENT: return "bad parent attribute";
    case DMI_ERR_ZERO_NORMAL: return "zero-length normal";
    case DMI_ERR_ENTROPY: return "entropy coder error";
    case DMI_ERR_ALPHABET_TOO_LARGE: return "symbol alphabet too large";
    case DMI_ERR_NO_DEVICE: return "no HIP device (libdraco_mi has no CPU fallback)";
    case DMI_ERR_HIP: return "HIP runtime error";
    case DMI_ERR_OUT_OF_MEMORY: return "out of device memory";
    case DMI_ERR_CONNECTIVITY: return "connectivity encoding error";
    case DMI_ERR_UNUSED_VERTICES: return "mesh contains unused vertices";
    case DMI_ERR_IO: return "i/o error";
  }
  return "unknown";
}
const char* dmi_last_error(void) { return g_last_error.c_str(); }
int dmi_last_call_timings(dmi_timings* t) {
  if (!t) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  *t = g_last_call;
  return DMI_OK;
}

void dmi_thread_host_threads(uint32_t n) { g_thread_host_cap = n; }
int dmi_usable_host_threads(void) { return (int)host_threads(); }

int dmi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// What the library keeps between calls so that the next one does not pay for it again: released device chunks (hipMalloc / hipFree
// serialise), idle pinned staging buffers, and the large host arrays of the connectivity stage (dmi_host.hpp).  Live jobs are untouched.
extern "C++" { namespace dmi { void gltf_pool_drop_all(); void host_blocks_drop_parked(); } }   // dmi_gltf.cpp (output arena blocks), dmi_hostmem.cpp (parked dmi_host_alloc blocks)
void dmi_release_cached_memory(void) {
  host_pool_drop_all();
  gltf_pool_drop_all();
  host_blocks_drop_parked();
  g_chunk_cache.drop_all();
  std::lock_guard<std::mutex> lock(g_stage_mutex);
  for (HostStage* st : g_stages) if (!st->in_use && st->p) stage_free(st);
}

void dmi_free(dmi_buffer* b) {
  if (!b) return;
  std::free(b->data);
  b->data = nullptr;
  b->len = b->cap = 0;
}
void dmi_free_many(dmi_buffer* bufs, uint32_t n) {
  if (!bufs) return;
  for (uint32_t k = 0; k < n; ++k) dmi_free(&bufs[k]);
}

}  // extern "C"
namespace dmi {
int to_buffer(const std::vector<uint8_t>& v, dmi_buffer* out) {
  out->data = static_cast<uint8_t*>(std::malloc(v.size() ? v.size() : 1));
  if (!out->data) return fail(DMI_ERR_OUT_OF_MEMORY, "malloc");
  if (!v.empty()) std::memcpy(out->data, v.data(), v.size());
  out->len = out->cap = v.size();
  return DMI_OK;
}
}  // namespace dmi
// The early stage of a whole-mesh call (EarlyQuant, dmi_job.hpp).  Only for the attribute set of a fused sweep in its packed layouts — a Position
// (coordinate-wise, 3 components, ≤ 21 bits) with at most one Normal (octahedral) and one TextureCoordinate (2 components, ≤ 16 bits), values in
// device memory — out stays null for anything else.  Whether the job will really run that sweep is only known once its corner tables are (an
// attribute with seams of its own leaves the sweep): job_create_impl compares its plan with this one.
int dmi::early_quantize_issue(const dmi_attribute* atts, uint32_t n_atts, const dmi_config& cfg, hipStream_t side, std::unique_ptr<EarlyQuant>& out) {
  out.reset();
  if (!atts || n_atts == 0 || n_atts > 3 || !side || dbg_on(DMI_DBG_NO_EARLY | DMI_DBG_NO_PACKED | DMI_DBG_NO_FUSED)) return DMI_OK;
  std::vector<AttJob> plan;
  if (validate_and_plan(atts, n_atts, cfg, plan) != DMI_OK) return DMI_OK;   // (the call itself reports the error)
  int i_pos = -1, i_nrm = -1, i_uv = -1;
  std::unique_ptr<EarlyQuant> e(new EarlyQuant());
  e->device = cfg.device; e->stream = side;
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = plan[i];
    const dmi_attribute& d = atts[i];
    EarlyQuant::Att ea{d.values, d.num_unique, (int)d.num_components, a.port == kCoordwise ? 0 : 1, QF_I32, a.bits, nullptr};
    // per-point attributes only (one record per value holds all of them), values in device memory
    if (!d.values || !d.num_unique || d.component_type != DMI_F32 || d.point_to_value || d.num_unique != atts[0].num_unique) return DMI_OK;
    if (d.att_type == DMI_ATT_POSITION && a.port == kCoordwise && d.num_components == 3 && a.bits <= 21 && a.scheme == kParallelogram && i_pos < 0) { ea.fmt = QF_P64; i_pos = (int)i; }
    else if (d.att_type == DMI_ATT_NORMAL && a.port == kOct && d.num_components == 3 && i_nrm < 0) { ea.fmt = QF_B16; i_nrm = (int)i; }
    else if (d.att_type == DMI_ATT_TEXCOORD && a.port == kCoordwise && d.num_components == 2 && a.bits <= 16 && i_uv < 0) { ea.fmt = QF_H32; i_uv = (int)i; }
    else return DMI_OK;
    e->atts.push_back(ea);
  }
  if (i_pos != 0 || (i_nrm < 0 && i_uv < 0)) return DMI_OK;   // (positions alone are no fused sweep)
  const uint32_t n = atts[0].num_unique;
  e->mem.init(cfg.device, side, (size_t)n * 16 + e->atts.size() * ((size_t)kRangeMaxBlocks * 8 * 4 + (size_t)value_quantize_rec_blocks(n) * 8 + 8192) + ((size_t)1 << 16));
  e->rec = e->mem.take<uint8_t>((size_t)n * 16);
  if (!e->rec) return DMI_OK;   // (no memory to spare: the job quantizes as always)
  RangeArgs ra{};
  for (size_t i = 0; i < e->atts.size(); ++i) {
    EarlyQuant::Att& ea = e->atts[i];
    ea.slot = e->mem.take<uint8_t>(256);
    float* partials = e->mem.take<float>((size_t)kRangeMaxBlocks * 8);
    if (!ea.slot || !partials) return DMI_OK;
    RangeAtt& r = ra.a[ra.count++];
    r.raw = static_cast<const float*>(ea.values);
    r.partials = partials;
    r.small = reinterpret_cast<uint32_t*>(ea.slot);         // the slot of a job's slab: [small 64 B][meta 64 B] — written by the first block of k_value_quantize_rec
    r.meta = reinterpret_cast<float*>(ea.slot + 64);
    r.zero = reinterpret_cast<uint32_t*>(ea.slot + 64);
    r.zero_words = 16;
    r.n = ea.n; r.N = ea.N; r.kind = ea.kind == 1 ? 2 : ea.kind;   // (normals: no range pass — the zero-length check rides in the quantizer, which reads them anyway)
  }
  HIP_TRY(hipEventCreate(&e->t0)); HIP_TRY(hipEventCreate(&e->t1));
  HIP_TRY(hipEventRecord(e->t0, side));
  launch_value_range_partials(ra, kEarlyRangeBlocks, side);   // (no `_final`: every block of the quantizer folds the pairs itself)
  ValueRecArgs va{};
  const int order[3] = {i_pos, i_nrm, i_uv};
  va.pos = static_cast<const float*>(e->atts[(size_t)i_pos].values);
  va.pos_partials = ra.a[i_pos].partials;
  va.pos_maxq = (float)(uint64_t)((1ull << e->atts[(size_t)i_pos].bits) - 1ull);
  if (i_nrm >= 0) { va.nrm = static_cast<const float*>(e->atts[(size_t)i_nrm].values); va.nrm_flags = reinterpret_cast<uint32_t*>(ra.a[i_nrm].partials); e->nrm_flags = va.nrm_flags; }   // (kRangeMaxBlocks · 8 words: the quantizer has at most 2048 blocks)
  if (i_uv >= 0) {
    va.uv = static_cast<const float*>(e->atts[(size_t)i_uv].values);
    va.uv_partials = ra.a[i_uv].partials;
    va.uv_maxq = (float)(uint64_t)((1ull << e->atts[(size_t)i_uv].bits) - 1ull);
  }
  va.n = n; va.rec = e->rec;
  // the joint i32 min/max of each attribute's quantized values (the wrapped difference's, wrapped_difference.rs:36-52) come out of the same pass as
  // partial pairs per block; the first block of the kernel that consumes the records folds them into the job's slot (EarlySlots)
  e->ipartial_blocks = value_quantize_rec_blocks(n);
  for (int k = 0; k < 3; ++k) {
    e->att_of_kind[k] = order[k];
    if (order[k] < 0) continue;
    va.range_blocks[k] = ra.a[order[k]].blocks;
    va.slot[k] = reinterpret_cast<uint32_t*>(e->atts[(size_t)order[k]].slot);
    int32_t* ip = e->mem.take<int32_t>((size_t)e->ipartial_blocks * 2);
    if (!ip) return DMI_OK;
    va.ipartials[k] = ip;
    e->ipartials[k] = ip;
  }
  launch_value_quantize_rec(va, side);
  HIP_TRY(hipEventRecord(e->t1, side));
  out = std::move(e);
  return DMI_OK;
}

extern "C" {

int dmi_job_create(const dmi_attribute* atts, const dmi_corner_table* tables, uint32_t n_atts, const uint32_t* seeds, uint32_t n_seeds,
                   const dmi_config* cfg_in, dmi_job** job_out) {
  return dmi::job_create_impl(atts, tables, n_atts, seeds, n_seeds, cfg_in, nullptr, job_out, nullptr);
}
}  // extern "C"

// dev (nullable): the universal table's arrays as the device connectivity stage left them in HBM (dmi_prepare.cpp) — every table whose
// host arrays are tables[0]'s takes them from there instead of the host (no upload, no range check: the library built them).
int dmi::job_create_impl(const dmi_attribute* atts, const dmi_corner_table* tables, uint32_t n_atts, const uint32_t* seeds, uint32_t n_seeds,
                         const dmi_config* cfg_in, const DeviceTableView* dev, dmi_job** job_out, JobDefer* defer) {
  if (defer && !dev) return fail(DMI_ERR_INVALID_ARGUMENT, "deferred job creation needs device-resident tables");
  if (!atts || !tables || !job_out || n_atts == 0 || n_atts > 255) return fail(DMI_ERR_INVALID_ARGUMENT, "null argument or bad attribute count");
  dmi_config cfg{};
  if (cfg_in) cfg = *cfg_in;
  DebugScope debug_scope(cfg.debug);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  HIP_TRY(hipSetDevice(cfg.device));
  const auto t_enter = std::chrono::steady_clock::now();
  std::unique_ptr<dmi_job> job(new dmi_job());
  job->debug = dbg();
  cfg.debug = &job->debug;   // (the caller's struct need not outlive the call)
  job->cfg = cfg;
  if (cfg.stream) job->stream = static_cast<hipStream_t>(cfg.stream);
  else if (g_adopt_stream) { job->stream_owner = g_adopt_stream; job->stream = g_adopt_stream->s; }
  else {
    job->stream_owner = std::make_shared<StreamHolder>();
    HIP_TRY(hipStreamCreate(&job->stream_owner->s));
    job->stream = job->stream_owner->s;
  }
  hipStream_t s = job->stream;
  int rc = validate_and_plan(atts, n_atts, cfg, job->atts);
  if (rc) return rc;
  {   // device memory of the job: a pool sized from the mesh (tables 24 B/face + 88 B/vertex each, ≈ 31 B per coded component, raw values)
    const size_t F0 = tables[0].num_faces;
    size_t est = 0;
    for (uint32_t i = 0; i < n_atts; ++i) {
      const size_t V0 = tables[i].num_vertices;
      est += F0 * 24 + V0 * 88 + (size_t)atts[i].num_unique * atts[i].num_components * 4 + V0 * ((size_t)job->atts[i].nq * 31 + 49) + ((size_t)1 << 20);
    }
    job->pool.stream = defer ? defer->stream : s;   // (a batch clears the chunk on the coordinator's stream: its kernels follow on the same stream)
    job->pool.device = cfg.device;
    job->pool.chunk_bytes = est + est / 8;
    // A batch job's chunk is small and one memset clears it (a thousand jobs must not issue five memsets each).  A single job's chunk is
    // gigabytes (4.5 GB per 10M triangles: 0.9 ms of fill per call), of which only a few KB have to start as zeros: those are cleared
    // one by one below (`needs_clear`).  DMI_POISON=1 fills such a chunk with 0xA5 instead — the tests run once that way, so that no buffer
    // silently depends on what the chunk held.
    // Round 5: a batch job's chunk is not cleared either — a thousand jobs issued a thousand fills of their whole chunks on the coordinator's stream
    // (17 ms of device time per 1024-file transcode, in front of the relabelling); the few ranges that must start as zeros are RECORDED
    // (JobDefer::clears) and the coordinator clears them all in one launch.
    job->pool.zero = dbg_on(DMI_DBG_ZERO_CHUNKS);
    job->pool.poison = !job->pool.zero && dbg_on(DMI_DBG_POISON);
  }
  g_active_pool = dbg_on(DMI_DBG_NO_POOL) ? nullptr : &job->pool;
  const bool needs_clear = g_active_pool && !job->pool.zero;   // pooled buffers do not start as zeros: the ones that must are cleared where they are allocated
  // (a single job: the ranges are collected and cleared by ONE launch at the end of job creation — nothing reads them before the encode; round 5 issued a
  //  hipMemsetAsync each: eleven launches of ≈ 5 µs back to back on the job's stream)
  ClearRanges pending_clears{};
  auto flush_clears = [&](hipStream_t st) { launch_clear_ranges(pending_clears, st); pending_clears.count = 0; };
  auto clear_range = [&](void* p, size_t bytes, hipStream_t st) -> hipError_t {   // at the end of job creation, or by the batch's coordinator
    if (defer && needs_clear) { defer->clears.push_back(JobDefer::Clear{p, bytes}); return hipSuccess; }
    if (defer || (bytes & 3) || !bytes) return bytes ? hipMemsetAsync(p, 0, bytes, st) : hipSuccess;
    if (pending_clears.count == kClearRanges) flush_clears(st);
    pending_clears.add(p, bytes);
    return hipSuccess;
  };
  struct PoolGuard { ~PoolGuard() { g_active_pool = nullptr; } } pool_guard;

  const bool trace_create = dbg_on(DMI_DBG_TRACE);
  const auto tc0 = std::chrono::steady_clock::now();
  auto since_ms = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  const uint32_t F = tables[0].num_faces;
  const size_t C = (size_t)F * 3;
  for (uint32_t i = 0; i < n_atts; ++i) {
    if (tables[i].num_faces != F) return fail(DMI_ERR_INVALID_ARGUMENT, "all corner tables must have the same face count");
    if (!tables[i].corner_to_point || !tables[i].corner_to_vertex || !tables[i].opposite) return fail(DMI_ERR_INVALID_ARGUMENT, "corner table arrays missing");
  }
  if (seeds) for (uint32_t k = 0; k < n_seeds; ++k) if (seeds[k] >= C) return fail(DMI_ERR_INVALID_ARGUMENT, "seed corner outside [0, 3F)");
  // ---- resident layout of the connectivity inputs --------------------------------------------------------
  // The tables arrive in the mesh's own face/vertex numbering.  Every predictor walks them in the coding
  // (Edgebreaker) order, so they are re-indexed once, here, into that order (a pure relabelling: the
  // bitstream does not depend on internal corner / vertex ids):
  //   * vertices → their sequence index: c2r[c] = rank of vertex(c) in the table's sequence (NONE if never
  //     coded), so "already coded" (the reference's `contains`) is `c2r[c] < i` with no indirection;
  //   * faces → ordered by the smallest sequence index among their universal vertices (counting sort), so
  //     consecutive sequence entries touch consecutive corners;
  //   * corner-indexed arrays (c2p, opp, c2r) and corner ids stored in `seq` / `opp` follow the new face order.
  // Tables identical to an earlier one (seam-free attribute tables) reuse its device copies.
  job->tables.resize(n_atts);
  std::vector<std::vector<uint32_t>> host_seq(n_atts);
  std::vector<const uint32_t*> seq_of(n_atts, nullptr);
  for (uint32_t i = 0; i < n_atts; ++i) {
    TableDev& t = job->tables[i];
    t.F = F;
    t.V = tables[i].num_vertices;
    for (uint32_t j = 0; j < i && t.alias_of < 0; ++j) {
      if (job->tables[j].alias_of >= 0) continue;
      if (tables[j].num_vertices != t.V) continue;
      const bool same_ptr = tables[j].corner_to_vertex == tables[i].corner_to_vertex && tables[j].opposite == tables[i].opposite;
      bool same = same_ptr;
      if (!same) {   // contents (a caller that keeps one table object per attribute): compared in parallel slices, giving up at the first difference
        std::atomic<int> differ{0};
        const uint32_t* a0 = tables[j].corner_to_vertex; const uint32_t* a1 = tables[i].corner_to_vertex;
        const uint32_t* b0 = tables[j].opposite; const uint32_t* b1 = tables[i].opposite;
        parallel_for(C, [&](size_t lo, size_t hi) {
          constexpr size_t kStep = 1u << 16;
          for (size_t at = lo; at < hi && !differ.load(std::memory_order_relaxed); at += kStep) {
            const size_t n = std::min(kStep, hi - at);
            if (std::memcmp(a0 + at, a1 + at, n * 4) != 0 || std::memcmp(b0 + at, b1 + at, n * 4) != 0) differ.store(1, std::memory_order_relaxed);
          }
        });
        same = !differ.load();
      }
      if (same) t.alias_of = (int)j;
    }
    if (t.alias_of >= 0) { job->atts[i].table = t.alias_of; continue; }
    const bool from_device = dev && tables[i].corner_to_vertex == tables[0].corner_to_vertex && tables[i].opposite == tables[0].opposite;
    // caller-supplied tables index host and device arrays below: every entry of a distinct table is range-checked once, here (error codes, not crashes).  A call with the
    // library's own device tables (dev->trusted_sequences: the prepare paths) built its seam tables itself, from those: three passes over 3F entries per seam table were
    // 5 % of a seam transcode's CPU samples
    if (!from_device && !(dev && dev->trusted_sequences)) {
      const uint32_t V = tables[i].num_vertices;
      const uint32_t* c2v = tables[i].corner_to_vertex;
      const uint32_t* opp = tables[i].opposite;
      const uint32_t* lmc = tables[i].left_most_corner;
      std::atomic<int> bad{0};
      parallel_for(C, [&](size_t lo, size_t hi) {
        int b = 0;
        for (size_t c = lo; c < hi; ++c) { if (c2v[c] >= V) b |= 1; if (opp[c] != kNone && opp[c] >= C) b |= 2; }
        if (b) bad.fetch_or(b);
      });
      if (lmc) parallel_for(V, [&](size_t lo, size_t hi) { for (size_t v = lo; v < hi; ++v) if (lmc[v] != kNone && lmc[v] >= C) { bad.fetch_or(4); break; } });
      if (bad & 1) return fail(DMI_ERR_INVALID_ARGUMENT, "corner table " + std::to_string(i) + ": corner_to_vertex entry ≥ num_vertices");
      if (bad & 2) return fail(DMI_ERR_INVALID_ARGUMENT, "corner table " + std::to_string(i) + ": opposite entry outside [0, 3F)");
      if (bad & 4) return fail(DMI_ERR_INVALID_ARGUMENT, "corner table " + std::to_string(i) + ": left_most_corner entry outside [0, 3F)");
    }
    const uint32_t* seq = tables[i].sequence;
    uint32_t n_seq = tables[i].sequence_len;
    if (!seq) {
      if (!seeds && n_seeds) return fail(DMI_ERR_INVALID_ARGUMENT, "no sequence and no seeds");
      if (!tables[i].left_most_corner) return fail(DMI_ERR_INVALID_ARGUMENT, "left_most_corner needed to compute the sequence");
      TableRef tr{F, t.V, tables[i].corner_to_vertex, tables[i].opposite, tables[i].left_most_corner};
      attribute_sequence(tr, seeds, n_seeds, host_seq[i]);
      seq = host_seq[i].data();
      n_seq = (uint32_t)host_seq[i].size();
    }
    if (!(from_device && dev->trusted_sequences)) {
      std::atomic<int> bad_seq{0};
      parallel_for(n_seq, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) if (seq[k] >= C) { bad_seq.store(1); break; } });
      if (bad_seq) return fail(DMI_ERR_INVALID_ARGUMENT, "sequence entry out of range");
    }
    t.n_seq = n_seq;
    seq_of[i] = seq;
  }
  const double t_seq = since_ms(tc0);
  // Large meshes are relabelled by kernels (dmi_relabel.hip): the caller's arrays go up as they are and the rank scatter, the face
  // keys, the stable sort and the corner remaps run on the device — what remains on the host is validation and the PCIe upload.
  // Small meshes (launch-bound: a batch creates thousands of jobs on host threads) keep the host form below.  Same arrays either way.
  // Host-relabel form: everything job creation uploads (relabelled tables, sequences, raw attribute values, composed maps) is written
  // by the host threads straight into ONE pinned staging buffer laid out like one region of the job's device memory, and goes up in a
  // single copy — a batch creates a thousand jobs on a hundred threads, and per-array copies from pageable memory serialise in the
  // runtime (57 ms of thread time per mesh before; 128 workers gained nothing over 16).
  struct StageGuard { HostStage* st = nullptr; ~StageGuard() { release_stage(st); } } stage_guard;
  uint8_t* stage_host = nullptr;
  uint8_t* stage_dev = nullptr;
  size_t stage_at = 0, stage_cap = 0;
  auto staged = [&](DevMem& m, size_t bytes) -> void* {   // a sub-array of the upload region; returns where the host writes it
    const size_t at = stage_at;
    stage_at += (bytes + 255) & ~(size_t)255;
    if (stage_at > stage_cap) return nullptr;
    m.p = stage_dev + at; m.bytes = bytes; m.pooled = true;   // (a view: the region owns the memory)
    return stage_host + at;
  };
  bool device_relabel = F >= kDeviceRelabelMinFaces || dev != nullptr;
  if (dbg().relabel == 1) device_relabel = true; else if (dbg().relabel == 2) device_relabel = dev != nullptr;
  // A one-shot call (create → encode → destroy) whose attributes ALL ride one fused sweep does not re-order its faces: the sweep reads fan rows, the
  // rows are built once from (seq, c2r, opp) in any numbering the three agree on, and the coding-order relabelling — a locality measure for the
  // per-attribute kernels that chase opp / c2r in every encode — would be paid (face keys, counting sort, table remap: 0.65 ms of kernels per 10M
  // faces) to be used exactly once, by the fan build.  The bitstream does not depend on internal face ids (tests: both forms against the oracle).
  bool plain_order = false;
  if (dev && dev->trusted_sequences && (defer || g_one_shot_call) && n_atts >= 2 && !dbg_on(DMI_DBG_NO_FUSED | DMI_DBG_NO_PLAIN_ORDER)) {   // (a batch's jobs are one-shot too)
    const AttJob& p = job->atts[0];
    int n_nrm = 0, n_uv = 0;
    bool all = p.scheme == kParallelogram && p.nq == 3 && job->tables[0].alias_of < 0 && p.table == 0;
    for (uint32_t i = 1; i < n_atts && all; ++i) {
      const AttJob& a = job->atts[i];
      all = (a.scheme == kNormal || a.scheme == kTexCoord) && a.parent == 0 && a.table == 0 && (a.scheme == kNormal ? ++n_nrm : ++n_uv) == 1;
    }
    plain_order = all;
  }
  TempDev tmpdev;
  {
    size_t hint = (size_t)64 << 10;
    for (uint32_t i = 0; i < n_atts; ++i) if (atts[i].point_to_value) hint += (size_t)atts[i].num_points * 4 + 256;
    uint32_t max_seq = 0, max_v = 0;
    for (uint32_t i = 0; i < n_atts; ++i) if (job->tables[i].alias_of < 0) { max_seq = std::max(max_seq, job->tables[i].n_seq); max_v = std::max(max_v, job->tables[i].V); }
    hint += C * 4 * 3 + ((size_t)max_seq + max_v) * 4 + (size_t)F * 4 * 9 + ((size_t)1 << 20);   // (sort scratch ≈ two key/value pairs)
    tmpdev.init(cfg.device, s, hint);
  }
  const bool faces_checked = dev && dev->trusted_sequences;   // tables built by dmi_conn.hip from faces it range-checked against the point count
  hipEvent_t ev_c0 = nullptr, ev_c1 = nullptr;   // device span of job creation (DMI_FLAG_TIMINGS, not for deferred batch jobs)
  struct EvGuard { hipEvent_t &a, &b; ~EvGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev_guard{ev_c0, ev_c1};
  if ((cfg.flags & DMI_FLAG_TIMINGS) && !defer) { HIP_TRY(hipEventCreate(&ev_c0)); HIP_TRY(hipEventCreate(&ev_c1)); HIP_TRY(hipEventRecord(ev_c0, s)); }
  uint32_t* d_bad = nullptr;        // device flag: a point_to_value entry out of range (device form)
  uint32_t* d_max_point = nullptr;  // device word: largest point index the faces reference (device form)
  if (defer) {
    // every distinct table becomes an item of the coordinator's batched relabelling: the universal one from the device arrays of the
    // connectivity stage, an attribute table of its own (interior seams) from the host arrays the walks used — the coordinator uploads those
    if (job->tables[0].alias_of >= 0) return fail(DMI_ERR_INVALID_ARGUMENT, "deferred job creation: table 0 must be the universal one");
    for (uint32_t i = 0; i < n_atts; ++i) {
      TableDev& t = job->tables[i];
      if (t.alias_of >= 0) continue;
      if ((rc = t.c2r.alloc(C * 4))) return rc;
      if ((rc = t.opp.alloc(C * 4))) return rc;
      if ((rc = t.seq.alloc((size_t)t.n_seq * 4))) return rc;
      if ((rc = t.s2p.alloc((size_t)t.n_seq * 4))) return rc;
      RelabelItem r{};
      bool resident = tables[i].corner_to_vertex == tables[0].corner_to_vertex && tables[i].opposite == tables[0].opposite;
      r.c2p = dev->c2p; r.c2v = resident ? dev->c2v : nullptr; r.opp = resident ? dev->opp : nullptr; r.seq = seq_of[i];
      for (uint32_t k = 0; k < dev->n_att && !resident; ++k)   // an attribute table the device built: its device copies
        if (dev->att_key[k] == tables[i].corner_to_vertex) { r.c2v = dev->att_c2v[k]; r.opp = dev->att_opp[k]; resident = true; }
      r.F = F; r.V = t.V; r.n_seq = t.n_seq; r.order_item = 0; r.plain = plain_order ? 1u : 0u;
      r.c2r = t.c2r.as<uint32_t>(); r.opp_out = t.opp.as<uint32_t>(); r.seq_out = t.seq.as<uint32_t>(); r.s2p = t.s2p.as<uint32_t>();
      defer->relabels.push_back(r);
      defer->host_c2v.push_back(resident ? nullptr : tables[i].corner_to_vertex);
      defer->host_opp.push_back(resident ? nullptr : tables[i].opposite);
    }
  } else if (device_relabel && plain_order) {
    // The mesh's own face order (see plain_order above): ranks, c2r = rank ∘ c2v, the device stage's opposite corners as they are, the sequence as the
    // host walk wrote it.  No face keys, no counting sort, no table remap: 0.4 ms of kernels instead of 1.05 for 10M faces.
    TableDev& t = job->tables[0];
    uint32_t* d_rank = tmpdev.take<uint32_t>(t.V ? t.V : 1);
    if (!d_rank) return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (relabelling temporaries)");
    // Round 6: ranks and opposite corners of a face in ONE 32-byte record (launch_face_records) instead of a corner → rank array beside `opp`: a swing of the
    // fan-row walk is then one read, not two on different lines (k_build_fans 346 → … µs per 10M faces); only overflow rows and deferred texture
    // coordinates read the records afterwards (FusedArgs::face_stride = 8)
    if ((rc = t.frec.alloc((size_t)F * 32))) return rc;
    if (dev->donor && dev->donor->device == cfg.device) {   // the stage's own array, in place: its pool's chunks now belong to the job
      t.opp.p = const_cast<uint32_t*>(dev->opp); t.opp.bytes = C * 4; t.opp.pooled = true;
      job->donated.device = dev->donor->device; job->donated.stream = s; job->donated.zero = false;
      job->donated.chunks.swap(dev->donor->chunks);
    }   // (no donor: the stage's array is read once, by launch_face_records below, on this same stream — the stage outlives job creation; no copy)
    // the sequence: already on the device when the call shipped it during the walk (SeqStream) — job creation waits for the last piece only
    if (g_seq_stream && !g_seq_stream->failed.load() && g_seq_stream->host == seq_of[0] && g_seq_stream->n == t.n_seq && g_seq_stream->device == cfg.device && t.n_seq) {
      HIP_TRY(hipStreamWaitEvent(s, g_seq_stream->ev, 0));
      t.seq.p = g_seq_stream->d_seq; t.seq.bytes = (size_t)t.n_seq * 4; t.seq.pooled = true;   // (a view: the stream object owns the memory, the job keeps it alive)
      job->seq_stream = g_seq_stream;
    } else {
      if ((rc = t.seq.alloc((size_t)t.n_seq * 4))) return rc;
      if (t.n_seq) HIP_TRY(hipMemcpyAsync(t.seq.p, seq_of[0], (size_t)t.n_seq * 4, hipMemcpyHostToDevice, s));
    }
    g_seq_stream.reset();
    if ((rc = t.s2p.alloc((size_t)t.n_seq * 4))) return rc;
    launch_fill_u32(d_rank, t.V, kNone, s);
    launch_rank_and_points(t.seq.as<uint32_t>(), t.n_seq, dev->c2v, dev->c2p, d_rank, t.s2p.as<uint32_t>(), s);
    launch_face_records(dev->c2v, d_rank, dev->opp, F, t.frec.as<uint32_t>(), s);
  } else if (device_relabel) {
    uint32_t max_seq = 0, max_v = 0;
    for (uint32_t i = 0; i < n_atts; ++i) if (job->tables[i].alias_of < 0) { max_seq = std::max(max_seq, job->tables[i].n_seq); max_v = std::max(max_v, job->tables[i].V); }
    bool host_tables = !dev;   // some table still comes from host arrays
    for (uint32_t i = 0; i < n_atts && dev; ++i)
      if (job->tables[i].alias_of < 0 && !(tables[i].corner_to_vertex == tables[0].corner_to_vertex && tables[i].opposite == tables[0].opposite)) host_tables = true;
    const uint32_t* d_c2p = dev ? dev->c2p : tmpdev.take<uint32_t>(C);
    uint32_t* d_c2v_up = host_tables ? tmpdev.take<uint32_t>(C) : nullptr;
    uint32_t* d_opp_up = host_tables ? tmpdev.take<uint32_t>(C) : nullptr;
    uint32_t* d_seq = tmpdev.take<uint32_t>(max_seq);
    uint32_t* d_rank = tmpdev.take<uint32_t>(max_v);
    uint32_t* d_key = tmpdev.take<uint32_t>(F);
    uint32_t* d_order = tmpdev.take<uint32_t>(F);
    uint32_t* d_new_face = tmpdev.take<uint32_t>(F);
    uint32_t* d_words = tmpdev.take<uint32_t>(4);
    const uint32_t n_keys = job->tables[0].n_seq + 1;   // key n_seq: faces none of whose vertices was coded sort last
    uint32_t* d_count = tmpdev.take<uint32_t>((size_t)n_keys + 1);
    uint32_t* d_fill = tmpdev.take<uint32_t>((size_t)n_keys + 1);
    uint32_t* d_parts = tmpdev.take<uint32_t>(scan_partials_words(n_keys + 1));
    if (!d_c2p || (host_tables && (!d_c2v_up || !d_opp_up)) || !d_seq || !d_rank || !d_key || !d_order || !d_new_face || !d_words || !d_count || !d_fill || !d_parts)
      return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (relabelling temporaries)");
    d_bad = d_words; d_max_point = d_words + 1;
    HIP_TRY(hipMemsetAsync(d_words, 0, 16, s));
    if (!dev) HIP_TRY(hipMemcpyAsync(const_cast<uint32_t*>(d_c2p), tables[0].corner_to_point, C * 4, hipMemcpyHostToDevice, s));
    if (!faces_checked) launch_max_u32(d_c2p, C, d_max_point, s);   // (the library's own connectivity stage range-checked the faces it built the tables from)
    for (uint32_t i = 0; i < n_atts; ++i) {
      TableDev& t = job->tables[i];
      if (t.alias_of >= 0) continue;
      const bool resident = dev && tables[i].corner_to_vertex == tables[0].corner_to_vertex && tables[i].opposite == tables[0].opposite;
      const uint32_t* d_c2v = resident ? dev->c2v : d_c2v_up;
      const uint32_t* d_opp = resident ? dev->opp : d_opp_up;
      if (!resident) {
        HIP_TRY(hipMemcpyAsync(d_c2v_up, tables[i].corner_to_vertex, C * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_opp_up, tables[i].opposite, C * 4, hipMemcpyHostToDevice, s));
      }
      if (t.n_seq) HIP_TRY(hipMemcpyAsync(d_seq, seq_of[i], (size_t)t.n_seq * 4, hipMemcpyHostToDevice, s));
      launch_fill_u32(d_rank, t.V, kNone, s);
      launch_rank_scatter(d_seq, t.n_seq, d_c2v, d_rank, s);
      if (i == 0) {   // the face order comes from the universal table
        HIP_TRY(launch_face_order(d_c2v, d_rank, F, n_keys, d_key, d_count, d_fill, d_parts, d_order, d_new_face, s));
      }
      if ((rc = t.c2r.alloc(C * 4))) return rc;
      if ((rc = t.opp.alloc(C * 4))) return rc;
      if ((rc = t.seq.alloc((size_t)t.n_seq * 4))) return rc;
      if ((rc = t.s2p.alloc((size_t)t.n_seq * 4))) return rc;
      launch_remap_table(d_c2v, d_opp, d_rank, d_order, d_new_face, C, t.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), s);
      launch_remap_seq(d_seq, t.n_seq, d_new_face, d_c2p, t.seq.as<uint32_t>(), t.s2p.as<uint32_t>(), s);
    }
  } else {
    // face order from table 0 (the universal table)
    // (scratch arrays are allocated uninitialised and first touched by the threads that fill them)
    auto raw_u32 = [](size_t n) { return std::unique_ptr<uint32_t[]>(new uint32_t[n ? n : 1]); };
    auto fill_none = [](uint32_t* p, size_t n) { parallel_for(n, [&](size_t lo, size_t hi) { std::fill(p + lo, p + hi, kNone); }); };
    auto rank0_buf = raw_u32(tables[0].num_vertices);
    uint32_t* rank0 = rank0_buf.get();
    fill_none(rank0, tables[0].num_vertices);
    parallel_for(job->tables[0].n_seq, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) rank0[tables[0].corner_to_vertex[seq_of[0][k]]] = (uint32_t)k; });
    const uint32_t nkeys = job->tables[0].n_seq + 1;   // key n_seq: faces none of whose vertices was coded
    auto key_buf = raw_u32(F);
    uint32_t* key = key_buf.get();
    std::vector<uint32_t> start(nkeys + 1, 0);
    parallel_for(F, [&](size_t lo, size_t hi) {
      for (size_t f = lo; f < hi; ++f) {
        uint32_t m = kNone;
        for (int k = 0; k < 3; ++k) m = std::min(m, rank0[tables[0].corner_to_vertex[3 * f + k]]);
        key[f] = (m == kNone) ? nkeys - 1 : m;
      }
    });
    for (uint32_t f = 0; f < F; ++f) ++start[key[f] + 1];
    for (uint32_t k = 0; k < nkeys; ++k) start[k + 1] += start[k];
    auto new_face_buf = raw_u32(F);
    uint32_t* new_face = new_face_buf.get();
    for (uint32_t f = 0; f < F; ++f) new_face[f] = start[key[f]]++;
    auto map_corner = [&](uint32_t c) { return c == kNone ? kNone : 3u * new_face[c / 3u] + c % 3u; };
    {   // the upload region and its pinned staging
      size_t need = 0;
      auto add = [&](size_t bytes) { need += (bytes + 255) & ~(size_t)255; };
      for (uint32_t i = 0; i < n_atts; ++i) { if (job->tables[i].alias_of < 0) { add(C * 4); add(C * 4); add((size_t)job->tables[i].n_seq * 4); add((size_t)job->tables[i].n_seq * 4); } }
      for (uint32_t i = 0; i < n_atts; ++i) { add((size_t)atts[i].num_unique * atts[i].num_components * 4); if (atts[i].point_to_value) add((size_t)job->tables[job->atts[i].table].n_seq * 4); }
      if ((rc = job->upload_region.alloc(need))) return rc;
      stage_guard.st = acquire_stage(cfg.device, need);
      if (!stage_guard.st) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (upload staging)");
      stage_host = stage_guard.st->p; stage_dev = job->upload_region.as<uint8_t>(); stage_cap = need;
    }
    std::unique_ptr<uint32_t[]> rank_buf;
    for (uint32_t i = 0; i < n_atts; ++i) {
      TableDev& t = job->tables[i];
      if (t.alias_of >= 0) continue;
      rank_buf = raw_u32(t.V);
      uint32_t* rank = rank_buf.get();
      fill_none(rank, t.V);
      parallel_for(t.n_seq, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) rank[tables[i].corner_to_vertex[seq_of[i][k]]] = (uint32_t)k; });
      uint32_t* tmp = static_cast<uint32_t*>(staged(t.c2r, C * 4));
      uint32_t* tmp2 = static_cast<uint32_t*>(staged(t.opp, C * 4));
      uint32_t* seq2 = static_cast<uint32_t*>(staged(t.seq, (size_t)t.n_seq * 4));
      uint32_t* s2p = static_cast<uint32_t*>(staged(t.s2p, (size_t)t.n_seq * 4));
      if (!tmp || !tmp2 || !seq2 || !s2p) return fail(DMI_ERR_HIP, "upload staging overflow");
      parallel_for(C, [&](size_t lo, size_t hi) {
        for (size_t c = lo; c < hi; ++c) {
          const uint32_t c2 = map_corner((uint32_t)c);
          tmp[c2] = rank[tables[i].corner_to_vertex[c]];
          tmp2[c2] = map_corner(tables[i].opposite[c]);
        }
      });
      // the corners of the sequence in the new face order, and the point every sequence entry stands for
      // (attribute_encoder.rs:332-338 reads attribute.get(point_idx(c)))
      parallel_for(t.n_seq, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) { seq2[k] = map_corner(seq_of[i][k]); s2p[k] = tables[0].corner_to_point[seq_of[i][k]]; } });
      t.s2p_host = s2p;
    }
  }

  size_t stage_copied = 0;
  if (stage_host && stage_at) {   // the tables go up now (the fan-row kernels below read them), the attribute arrays after their loop
    HIP_TRY(hipMemcpyAsync(stage_dev, stage_host, stage_at, hipMemcpyHostToDevice, s));
    stage_copied = stage_at;
  }
  const double t_relabel = since_ms(tc0) - t_seq;
  // Seam-free fast path: a normal / texture-coordinate attribute coded on the same corner table as its parent
  // position attribute (3 components, parallelogram) is predicted together with it in one sweep (k_predict_fused).
  if (!dbg_on(DMI_DBG_NO_FUSED)) {
    for (uint32_t i = 0; i < n_atts; ++i) {
      AttJob& a = job->atts[i];
      if ((a.scheme != kNormal && a.scheme != kTexCoord) || a.parent < 0) continue;
      AttJob& p = job->atts[a.parent];
      if (p.scheme != kParallelogram || p.nq != 3 || p.table != a.table) continue;
      int& slot = a.scheme == kNormal ? p.fused_nrm : p.fused_uv;
      if (slot >= 0) continue;   // one of each kind per sweep; further ones take their own kernels
      slot = (int)i;
      a.fused_into = a.parent;
    }
  }
  // Packed quantized values for the attributes of a fused sweep (QFmt: positions ≤ 21 bits in one uint64, octahedral normals in a
  // uint16, texture coordinates ≤ 16 bits in a uint32) — all three of a sweep or none; symbols as uint16 where the alphabet bound allows.
  if (!dbg_on(DMI_DBG_NO_PACKED)) {
    for (auto& a : job->atts) {
      if (a.fused_nrm < 0 && a.fused_uv < 0) continue;
      if (a.port != kCoordwise || a.bits > 21) continue;
      if (a.fused_uv >= 0 && (job->atts[a.fused_uv].port != kCoordwise || job->atts[a.fused_uv].bits > 16)) continue;
      a.qfmt = QF_P64;
      if (a.fused_nrm >= 0) job->atts[a.fused_nrm].qfmt = QF_B16;
      if (a.fused_uv >= 0) job->atts[a.fused_uv].qfmt = QF_H32;
    }
  }
  if (!dbg_on(DMI_DBG_NO_SYM16)) for (auto& a : job->atts) a.sym16 = a.port != kToBits && symbol_bins(a) <= 65536u;
  // the early stage of a whole-mesh call (EarlyQuant): adopted when this plan is the one it guessed — same values, same packed layouts, same bits
  if (g_early_quant && !defer && dev && dev->values_on_device && g_early_quant->device == cfg.device && g_early_quant->atts.size() == n_atts) {
    bool same = true;
    for (uint32_t i = 0; i < n_atts && same; ++i) {
      const EarlyQuant::Att& ea = g_early_quant->atts[i];
      const AttJob& a = job->atts[i];
      same = ea.values == a.desc.values && ea.n == a.desc.num_unique && !a.desc.point_to_value && a.table == job->atts[0].table && job->tables[a.table].n_seq == ea.n /* the sequence holds every value once: the stage's min/max are the sequence's */ && ea.fmt == a.qfmt && ea.fmt != QF_I32 && ea.bits == a.bits && ea.kind == (a.port == kCoordwise ? 0 : 1);
    }
    if (same) job->early = std::shared_ptr<EarlyQuant>(g_early_quant.release());
  }
  g_early_quant.reset();   // (not this job's plan: the early kernels finish on their stream and their memory goes back to the pool)
  for (auto& a : job->atts) {   // fan rows of the tables a fused sweep runs on: the corner table re-laid out per coded vertex
    if (a.fused_nrm < 0 && a.fused_uv < 0) continue;
    TableDev& t = job->tables[a.table];
    if (t.fan.p || t.n_seq == 0) continue;
    if ((rc = t.fan_hdr.alloc((size_t)t.n_seq * 4))) return rc;
    if ((rc = t.fan_apex.alloc((size_t)t.n_seq * 4))) return rc;
    if ((rc = t.fan.alloc((size_t)t.n_seq * 32))) return rc;
    if (defer) defer->fans.push_back(FanItem{t.seq.as<uint32_t>(), t.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), t.fan_hdr.as<uint32_t>(), t.fan_apex.as<uint32_t>(), t.fan.as<uint32_t>(), 0u, t.n_seq, 0u, 0u});
    else if (t.frec.p) launch_build_fans_rec(t.seq.as<uint32_t>(), t.n_seq, t.frec.as<uint32_t>(), t.fan_hdr.as<uint32_t>(), t.fan_apex.as<uint32_t>(), t.fan.as<uint32_t>(), s);
    else launch_build_fans(t.seq.as<uint32_t>(), t.n_seq, t.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), t.fan_hdr.as<uint32_t>(), t.fan_apex.as<uint32_t>(), t.fan.as<uint32_t>(), false, s);
  }
  uint32_t max_point = 0;
  if (defer) {
    // (the connectivity stage checked the faces against the points, its caller the attributes' point and value counts)
  } else if (device_relabel && faces_checked) {
    // (no read-back, no wait in the middle of job creation: everything up to the final synchronisation is queued behind the relabelling)
  } else if (device_relabel) {
    HIP_TRY(hipMemcpyAsync(&max_point, d_max_point, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else {
    std::atomic<uint32_t> mp{0};
    parallel_for(C, [&](size_t lo, size_t hi) {
      uint32_t m = 0;
      for (size_t c = lo; c < hi; ++c) m = std::max(m, tables[0].corner_to_point[c]);
      uint32_t cur = mp.load();
      while (m > cur && !mp.compare_exchange_weak(cur, m)) {}
    });
    max_point = mp.load();
  }
  size_t pinned_need = 0;
  uint64_t pb = 0;
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    const dmi_attribute& d = a.desc;
    const TableDev& t = job->tables[a.table];
    const size_t vbytes = (size_t)d.num_unique * d.num_components * 4;
    if (d.num_unique && !d.values && !(defer && i < defer->values_dev.size() && defer->values_dev[i])) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute values missing");
    if (defer) {
      if ((rc = a.raw.alloc(vbytes))) return rc;
      if (vbytes && i < defer->values_dev.size() && defer->values_dev[i]) defer->copies.push_back({a.raw.p, defer->values_dev[i], vbytes});
      else if (vbytes) HIP_TRY(hipMemcpyAsync(a.raw.p, d.values, vbytes, hipMemcpyHostToDevice, defer->stream));
    } else if (stage_host) {
      void* dst = staged(a.raw, vbytes);
      if (!dst) return fail(DMI_ERR_HIP, "upload staging overflow");
      if (vbytes >= ((size_t)8 << 20)) parallel_for(vbytes, [&](size_t lo, size_t hi) { std::memcpy(static_cast<uint8_t*>(dst) + lo, static_cast<const uint8_t*>(d.values) + lo, hi - lo); });
      else if (vbytes) std::memcpy(dst, d.values, vbytes);
    } else if (job->early) {
      // (no copy of the raw values: ranges and quantization ran on the caller's arrays, the pass gathers the packed values)
    } else {
      rc = upload(a.raw, d.values, vbytes, s, dev && dev->values_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
      if (rc) return rc;
    }
    if (d.num_points <= max_point && F) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + " has fewer points than the faces reference");
    if (d.point_to_value && defer) {
      if (!(i < defer->maps_dev.size() && defer->maps_dev[i])) return fail(DMI_ERR_INVALID_ARGUMENT, "deferred job creation: point_to_value map not resident");
      if ((rc = a.s2v.alloc((size_t)t.n_seq * 4))) return rc;
      defer->compose.push_back(ComposeItem{t.s2p.as<uint32_t>(), defer->maps_dev[i], a.s2v.as<uint32_t>(), 0u, t.n_seq});
    } else if (d.point_to_value && device_relabel) {   // sequence index → value index, composed on the device; out-of-range entries raise d_bad
      uint32_t* d_p2v = tmpdev.take<uint32_t>(d.num_points);
      if (!d_p2v) return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (point_to_value upload)");
      HIP_TRY(hipMemcpyAsync(d_p2v, d.point_to_value, (size_t)d.num_points * 4, hipMemcpyHostToDevice, s));
      if ((rc = a.s2v.alloc((size_t)t.n_seq * 4))) return rc;
      launch_compose_s2v(t.s2p.as<uint32_t>(), t.n_seq, d_p2v, d.num_points, d.num_unique, a.s2v.as<uint32_t>(), d_bad, s);
    } else if (d.point_to_value) {   // sequence index → value index: the map composed with the table's sequence → point array
      uint32_t* s2v = static_cast<uint32_t*>(staged(a.s2v, (size_t)t.n_seq * 4));
      if (!s2v) return fail(DMI_ERR_HIP, "upload staging overflow");
      for (uint32_t k = 0; k < t.n_seq; ++k) {
        s2v[k] = d.point_to_value[t.s2p_host[k]];
        if (s2v[k] >= d.num_unique) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + ": point_to_value entry out of range");
      }
    } else if (d.num_unique <= max_point && F) {
      return fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + " has fewer values than the faces reference");
    }
    const uint32_t n = t.n_seq;
    a.n_sym = (uint64_t)n * a.nq;
    if ((rc = a.qs.alloc(a.qfmt == QF_P64 ? (size_t)n * 8 : a.qfmt == QF_B16 ? (size_t)n * 2 + 2 : a.qfmt == QF_H32 ? (size_t)n * 4 : (size_t)n * a.nq * 4))) return rc;
    if ((rc = a.sym.alloc((size_t)a.n_sym * (a.sym16 ? 2 : 4) + 4))) return rc;
    if (a.scheme == kNormal || a.scheme == kTexCoord) {
      if ((rc = a.aux.alloc(n ? n : 1))) return rc;
      a.aux_cap = (uint64_t)n + 16;   // ≤ 1 byte per coded bit + flush
      if ((rc = a.aux_out.alloc(a.aux_cap + 16))) return rc;   // +16: the batch pack kernel copies whole 16-byte words
      if ((rc = a.aux_rec.alloc(((size_t)n + kChainPad) * sizeof(RansEntry)))) return rc;
      if (!a.aux_rec.pooled) HIP_TRY(hipMemsetAsync(a.aux_rec.p, 0, a.aux_rec.bytes, s));
      else if (needs_clear) HIP_TRY(clear_range(a.aux_rec.as<RansEntry>() + n, kChainPad * sizeof(RansEntry), s));   // (the records past n: the chains read ahead into them)
      if ((rc = a.aux_flags.alloc(((size_t)n / 64 + 4) * 4))) return rc;
      if (!a.aux_flags.pooled || needs_clear) HIP_TRY(clear_range(a.aux_flags.p, a.aux_flags.bytes, s));
      if ((rc = a.chunk_info.alloc((size_t)std::max(1u, orient_summary_blocks(n)) * 8 + 16))) return rc;
      if (a.scheme == kTexCoord && (rc = a.aux_bits.alloc((size_t)n + 16))) return rc;
      if (a.scheme == kNormal && (rc = a.flip_partials.alloc((size_t)kSweepMaxBlocks * 4))) return rc;
      if (a.scheme == kTexCoord && a.fused_into >= 0 && (rc = a.fix_list.alloc((size_t)n * 4 + 4))) return rc;
    }
    if (a.scheme == kNormal && a.fused_into < 0 && n) {   // fan rows: this table's fans, ranks in the parent position table
      const TableDev& pt = job->tables[job->atts[a.parent].table];
      if ((rc = a.fan_hdr.alloc((size_t)n * 4))) return rc;
      if ((rc = a.fan_apex.alloc((size_t)n * 4))) return rc;
      if ((rc = a.fan.alloc((size_t)n * 32))) return rc;
      if (defer) defer->fans.push_back(FanItem{t.seq.as<uint32_t>(), pt.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), a.fan_hdr.as<uint32_t>(), a.fan_apex.as<uint32_t>(), a.fan.as<uint32_t>(), 0u, n, 1u, 0u});
      else launch_build_fans(t.seq.as<uint32_t>(), n, pt.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), a.fan_hdr.as<uint32_t>(), a.fan_apex.as<uint32_t>(), a.fan.as<uint32_t>(), true, s);
    }
    a.bins = symbol_bins(a);
    if (a.port == kToBits) a.bins = 1u << 20;   // capacity; the real bound is checked after the min/max readback
    a.bins_cap = a.bins;
    if ((rc = a.rtable.alloc(((size_t)a.bins + 4) * sizeof(RansEntry)))) return rc;   // +4: uploads are padded to 4 entries (whole 16-byte words)
    if (a.port != kToBits) {
      a.hdr_cap = 8u + 3u * a.bins;   // method, bit_length, leb128(num_symbols), ≤ 3 bytes per symbol
      if ((rc = a.freq.alloc(((size_t)a.bins + 68) * 4))) return rc;
      if ((rc = a.hdr.alloc((size_t)a.hdr_cap + 32))) return rc;   // (+ slack: the batch pack kernel copies whole 16-byte words)
      if ((rc = a.aux_entries.alloc(64))) return rc;
    }
    if ((rc = a.rec.alloc(((size_t)a.n_sym + kChainPad) * sizeof(RansEntry)))) return rc;
    if (!a.rec.pooled) HIP_TRY(hipMemsetAsync(a.rec.p, 0, a.rec.bytes, s));
    else if (needs_clear) HIP_TRY(clear_range(a.rec.as<RansEntry>() + a.n_sym, kChainPad * sizeof(RansEntry), s));
    if ((rc = a.batch_flags.alloc(((size_t)a.n_sym / 64 + 4) * 4))) return rc;
    if (!a.batch_flags.pooled || needs_clear) HIP_TRY(clear_range(a.batch_flags.p, a.batch_flags.bytes, s));
    a.out_cap = a.n_sym * 3 + 16;   // ≤ 3 renormalisation bytes per symbol (P ≤ 20) + flush
    if ((rc = a.out.alloc(a.out_cap + 16))) return rc;
    if ((rc = a.partials.alloc((size_t)kRangeMaxBlocks * 8 * 4))) return rc;
    if ((rc = a.ipartials.alloc((size_t)kSeqQuantizeMaxBlocks * 2 * 4))) return rc;   // ≥ 2 * seq_quantize_blocks(n)
    // slab slot of this attribute: [small 64 B][meta 64 B][hist bins_cap·4][summary], 256-byte aligned
    a.slab_off = pinned_need;
    pinned_need += 128 + (size_t)a.bins_cap * 4 + (a.scheme == kTexCoord ? (size_t)std::max(1u, orient_summary_blocks(n)) * 16 : 0);
    pinned_need = (pinned_need + 255) & ~(size_t)255;
    // algorithmic bytes of the quantize+predict pass: 4·Nin + 4·Nsym per value, 8 per sequence entry
    pb += (uint64_t)d.num_unique * 4 * d.num_components + a.n_sym * 4 + (uint64_t)n * 8;
  }
  if (stage_host && stage_at > stage_copied) HIP_TRY(hipMemcpyAsync(stage_dev + stage_copied, stage_host + stage_copied, stage_at - stage_copied, hipMemcpyHostToDevice, s));
  pb += (uint64_t)F * 24;   // corner_to_point + opposite, once
  job->predict_bytes = pb;
  pinned_need += 256;
  if ((rc = job->slab.alloc(pinned_need))) return rc;
  if (!job->slab.pooled || needs_clear) HIP_TRY(clear_range(job->slab.p, pinned_need, s));
  for (auto& a : job->atts) {
    uint8_t* base = job->slab.as<uint8_t>() + a.slab_off;
    a.small = SlabView{base, 64};
    a.meta = SlabView{base + 64, 64};
    a.hist = SlabView{base + 128, (size_t)a.bins_cap * 4};
    a.summary = SlabView{base + 128 + (size_t)a.bins_cap * 4, a.scheme == kTexCoord ? (size_t)std::max(1u, orient_summary_blocks(job->tables[a.table].n_seq)) * 16 : 0};
  }
  job->pinned_bytes = pinned_need;   // (the pinned mirror is allocated by the first single-job encode: a batch reads back through its arena)
  job->dev_tables = !dbg_on(DMI_DBG_HOST_TABLES);
  for (auto& a : job->atts) if (a.port == kToBits) job->dev_tables = false;
  {   // where the serial coders of a single-job encode run: dmi_debug::chains forces, default = by the longest stream
    // (a host core steps ≈ 6× faster than a scalar-unit walker, but costs a read-back of the symbols and a few thread starts)
    uint64_t longest = 0;
    for (auto& a : job->atts) longest = std::max<uint64_t>(longest, a.n_sym);
    if (dbg().chains == 2) job->host_chains = true;
    else if (dbg().chains == 1) job->host_chains = false;
    else job->host_chains = longest >= kHostChainMinSymbols;
  }
  if ((rc = job->descs.alloc(sizeof(ChainDesc) * (size_t)n_atts * 2 + 16)   /* + the chain kernel's pull counter */)) return rc;
  if (cfg.flags & DMI_FLAG_TIMINGS) {
    for (auto& e : job->ev) HIP_TRY(hipEventCreate(&e));
    job->have_events = true;
  }
  // The quantize gather in tile-sorted order: inside tiles of consecutive sequence entries the slots are ordered by point index —
  // the gather's wavefronts then read neighbouring points.  Large single jobs only (a batch's meshes are smaller than a tile); sorted on the
  // device (launch_tile_sort: tiles of up to 16 K entries in one workgroup's LDS, larger ones with the long strides in global memory).  A tile has
  // to span about two rings of the coding order to pay (below).  DMI_TILE_SORT=0 switches it off, =<entries> picks the tile (rounded up to a
  // power of two).
  {
    // (read per job creation, not once: the tests run small meshes through every form — DMI_TILE_SORT_MIN lowers the length it starts at,
    //  DMI_TILE_SORT_LOCAL the block size, so that small meshes reach the global-memory strides)
    const int env_tile = dbg().tile_sort == 0 ? -1 : (dbg().tile_sort < 0 ? 0 : dbg().tile_sort);   // (below: -1 default, 0 off, > 0 the tile)
    const uint32_t min_entries = dbg().tile_sort_min ? dbg().tile_sort_min : kTileSortMinEntries;
    uint32_t local_lg = kTileSortMaxLog2;
    if (const uint32_t e = dbg().tile_sort_local) { local_lg = 6; while ((1u << local_lg) < e && local_lg < kTileSortMaxLog2) ++local_lg; }
    // a create → encode → destroy call sorts (≈ 0.31 ms per 10M triangles) for ONE gather that the sort makes 14–18 µs faster: only resident
    // jobs (dmi_job_create / dmi_mesh_prepare: encoded many times) or an explicit DMI_TILE_SORT take it
    if (env_tile != 0 && !defer && !(g_one_shot_call && env_tile < 0)) for (auto& t : job->tables) {
      if (t.alias_of >= 0 || t.n_seq == 0 || t.n_seq < min_entries) continue;
      // ≈ 2.2 rings of the coding order (a ring of a grid-like mesh is ≈ 4·√V entries), as a power of two between 16 K and 128 K — measured:
      // 10M triangles 16 K (32 K +2 %, 64 K +7 %), 20M 32 K (128 K +6 %), 40M 32–64 K (128 K +8 %), 100M 64–128 K (256 K +9 %)
      uint32_t lg = (uint32_t)std::min<long>(kTileSortBigLog2, std::max<long>(kTileSortMaxLog2, std::lround(std::log2(8.8 * std::sqrt((double)t.n_seq)))));
      if (env_tile > 0) { lg = 6; while ((1u << lg) < (uint32_t)env_tile && lg < 24) ++lg; }
      if ((rc = t.s2p_sorted.alloc((size_t)t.n_seq * 4)) || (rc = t.sorted_dest.alloc((size_t)t.n_seq * 4))) return rc;
      const size_t scratch_bytes = tile_sort_scratch_bytes(t.n_seq, lg, local_lg);
      uint64_t* scratch = nullptr;
      if (scratch_bytes) {
        scratch = tmpdev.take<uint64_t>(scratch_bytes / 8);
        if (!scratch) return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (tile sort keys)");
      }
      // (a layout optimisation the bitstream does not depend on: if its launch cannot be set up — 128 KB of LDS on a part that has less — the
      //  plain gather runs instead)
      if (launch_tile_sort(t.s2p.as<uint32_t>(), t.n_seq, lg, local_lg, scratch, t.s2p_sorted.as<uint32_t>(), t.sorted_dest.as<uint32_t>(), s) != hipSuccess) {
        (void)hipGetLastError();
        (void)t.s2p_sorted.alloc(0); (void)t.sorted_dest.alloc(0);
      }
    }
  }
  flush_clears(s);
  uint32_t bad_p2v = 0;
  if (d_bad) HIP_TRY(hipMemcpyAsync(&bad_p2v, d_bad, 4, hipMemcpyDeviceToHost, s));
  if (ev_c1) HIP_TRY(hipEventRecord(ev_c1, s));
  if (!defer) HIP_TRY(hipStreamSynchronize(s));   // (a batch waits once, for all its jobs)
  if (ev_c1) { float span = 0; if (hipEventElapsedTime(&span, ev_c0, ev_c1) == hipSuccess) job->create_device_ms = span; else (void)hipGetLastError(); }
  if (bad_p2v) return fail(DMI_ERR_INVALID_ARGUMENT, "point_to_value entry out of range");
  if (trace_create) std::fprintf(stderr, "[dmi] job create (%u faces, %s relabelling): sequences %.1f ms, relabel + table uploads %.1f, attribute uploads + buffers + fan rows %.1f, stream + plan %.1f\n", F, device_relabel ? "device" : "host", t_seq, t_relabel, since_ms(tc0) - t_seq - t_relabel,
                                 std::chrono::duration<double, std::milli>(tc0 - t_enter).count());
  *job_out = job.release();
  return DMI_OK;
}

extern "C" {
int dmi_tile_sort_slots(const uint32_t* sequence_to_point, uint32_t n, uint32_t tile_entries, uint32_t block_entries, const dmi_config* cfg_in, uint32_t* slot_point, uint32_t* slot_entry) {
  if ((!sequence_to_point || !slot_point || !slot_entry) && n) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  dmi_config cfg{};
  if (cfg_in) cfg = *cfg_in;
  DebugScope debug_scope(cfg.debug);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  HIP_TRY(hipSetDevice(cfg.device));
  if (!n) return DMI_OK;
  uint32_t lg = 6, local_lg = 6;
  while ((1u << lg) < tile_entries && lg < 24) ++lg;
  while ((1u << local_lg) < block_entries && local_lg < kTileSortMaxLog2) ++local_lg;
  struct Dev { void* p = nullptr; ~Dev() { if (p) (void)hipFree(p); } } in, outp, outd, scratch;
  HIP_TRY(hipMalloc(&in.p, (size_t)n * 4)); HIP_TRY(hipMalloc(&outp.p, (size_t)n * 4)); HIP_TRY(hipMalloc(&outd.p, (size_t)n * 4));
  const size_t sb = tile_sort_scratch_bytes(n, lg, local_lg);
  if (sb) HIP_TRY(hipMalloc(&scratch.p, sb));
  HIP_TRY(hipMemcpy(in.p, sequence_to_point, (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(launch_tile_sort(static_cast<const uint32_t*>(in.p), n, lg, local_lg, static_cast<uint64_t*>(scratch.p), static_cast<uint32_t*>(outp.p), static_cast<uint32_t*>(outd.p), nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(slot_point, outp.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(slot_entry, outd.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return DMI_OK;
}
void dmi_job_destroy(dmi_job* job) { delete job; }

int dmi_job_timings(const dmi_job* job, dmi_timings* t) {
  if (!job || !t) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  *t = job->last;
  return DMI_OK;
}

}  // extern "C"
