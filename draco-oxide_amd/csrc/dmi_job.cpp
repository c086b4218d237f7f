// dmi_job.cpp — C ABI of libdraco_mi.so (include/draco_mi.h) and the per-mesh job that drives the
// gfx950 kernels: upload once → value ranges → sequence-order gather + quantize → predict+transform → histograms →
// table stage (k_tables; host form behind DMI_HOST_TABLES) → record prep → walker/emitter rANS/rABS chains → byte splice;
// the batch drivers (one upload, one launch per kernel for all jobs, one chain launch, one packed read-back).
// There is NO CPU fallback: without a HIP device every encode entry point returns DMI_ERR_NO_DEVICE.
#include "dmi_job.hpp"

namespace dmi {
thread_local std::string g_last_error;
thread_local dmi_timings g_last_call{};
int host_fail(int code, const std::string& msg) { g_last_error = msg; return code; }   // shared with the host-only translation units
ChunkCache g_chunk_cache;
thread_local DevPool* g_active_pool = nullptr;
}  // namespace dmi

using namespace dmi;

thread_local std::shared_ptr<StreamHolder> g_adopt_stream;

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
  static const bool plain = std::getenv("DMI_NO_THP") != nullptr;
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

int dmi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// What the library keeps between calls so that the next one does not pay for it again: released device chunks (hipMalloc / hipFree
// serialise), idle pinned staging buffers, and the large host arrays of the connectivity stage (dmi_host.hpp).  Live jobs are untouched.
void dmi_release_cached_memory(void) {
  host_pool_drop_all();
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
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  HIP_TRY(hipSetDevice(cfg.device));
  const auto t_enter = std::chrono::steady_clock::now();
  std::unique_ptr<dmi_job> job(new dmi_job());
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
  }
  g_active_pool = std::getenv("DMI_NO_POOL") ? nullptr : &job->pool;
  struct PoolGuard { ~PoolGuard() { g_active_pool = nullptr; } } pool_guard;

  const bool trace_create = std::getenv("DMI_TRACE") != nullptr;
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
    if (!from_device) {   // caller-supplied tables index host and device arrays below: every entry of a distinct table is range-checked once, here (error codes, not crashes)
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
  if (const char* e = std::getenv("DMI_RELABEL")) device_relabel = std::strcmp(e, "device") == 0 ? true : (std::strcmp(e, "host") == 0 ? (dev != nullptr) : device_relabel);
  TempDev tmpdev;
  {
    size_t hint = (size_t)64 << 10;
    for (uint32_t i = 0; i < n_atts; ++i) if (atts[i].point_to_value) hint += (size_t)atts[i].num_points * 4 + 256;
    uint32_t max_seq = 0, max_v = 0;
    for (uint32_t i = 0; i < n_atts; ++i) if (job->tables[i].alias_of < 0) { max_seq = std::max(max_seq, job->tables[i].n_seq); max_v = std::max(max_v, job->tables[i].V); }
    hint += C * 4 * 3 + ((size_t)max_seq + max_v) * 4 + (size_t)F * 4 * 9 + ((size_t)1 << 20);   // (sort scratch ≈ two key/value pairs)
    tmpdev.init(cfg.device, s, hint);
  }
  uint32_t* d_bad = nullptr;        // device flag: a point_to_value entry out of range (device form)
  uint32_t* d_max_point = nullptr;  // device word: largest point index the faces reference (device form)
  if (defer) {
    for (uint32_t i = 1; i < n_atts; ++i) if (job->tables[i].alias_of != 0) return fail(DMI_ERR_INVALID_ARGUMENT, "deferred job creation: every table must be the universal one");
    TableDev& t = job->tables[0];
    if ((rc = t.c2r.alloc(C * 4))) return rc;
    if ((rc = t.opp.alloc(C * 4))) return rc;
    if ((rc = t.seq.alloc((size_t)t.n_seq * 4))) return rc;
    if ((rc = t.s2p.alloc((size_t)t.n_seq * 4))) return rc;
    defer->has_relabel = true;
    RelabelItem& r = defer->relabel;
    r.c2p = dev->c2p; r.c2v = dev->c2v; r.opp = dev->opp; r.seq = seq_of[0];
    r.F = F; r.V = t.V; r.n_seq = t.n_seq;
    r.c2r = t.c2r.as<uint32_t>(); r.opp_out = t.opp.as<uint32_t>(); r.seq_out = t.seq.as<uint32_t>(); r.s2p = t.s2p.as<uint32_t>();
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
    launch_max_u32(d_c2p, C, d_max_point, s);
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
  if (!std::getenv("DMI_NO_FUSED")) {
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
  if (!std::getenv("DMI_NO_PACKED")) {
    for (auto& a : job->atts) {
      if (a.fused_nrm < 0 && a.fused_uv < 0) continue;
      if (a.port != kCoordwise || a.bits > 21) continue;
      if (a.fused_uv >= 0 && (job->atts[a.fused_uv].port != kCoordwise || job->atts[a.fused_uv].bits > 16)) continue;
      a.qfmt = QF_P64;
      if (a.fused_nrm >= 0) job->atts[a.fused_nrm].qfmt = QF_B16;
      if (a.fused_uv >= 0) job->atts[a.fused_uv].qfmt = QF_H32;
    }
  }
  if (!std::getenv("DMI_NO_SYM16")) for (auto& a : job->atts) a.sym16 = a.port != kToBits && symbol_bins(a) <= 65536u;
  for (auto& a : job->atts) {   // fan rows of the tables a fused sweep runs on: the corner table re-laid out per coded vertex
    if (a.fused_nrm < 0 && a.fused_uv < 0) continue;
    TableDev& t = job->tables[a.table];
    if (t.fan.p || t.n_seq == 0) continue;
    if ((rc = t.fan_hdr.alloc((size_t)t.n_seq * 4))) return rc;
    if ((rc = t.fan_apex.alloc((size_t)t.n_seq * 4))) return rc;
    if ((rc = t.fan.alloc((size_t)t.n_seq * 32))) return rc;
    if (defer) defer->fans.push_back(FanItem{t.seq.as<uint32_t>(), t.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), t.fan_hdr.as<uint32_t>(), t.fan_apex.as<uint32_t>(), t.fan.as<uint32_t>(), 0u, t.n_seq, 0u, 0u});
    else launch_build_fans(t.seq.as<uint32_t>(), t.n_seq, t.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), t.fan_hdr.as<uint32_t>(), t.fan_apex.as<uint32_t>(), t.fan.as<uint32_t>(), false, s);
  }
  uint32_t max_point = 0;
  if (defer) {
    // (the connectivity stage checked the faces against the points, its caller the attributes' point and value counts)
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
    if (d.num_unique && !d.values) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute values missing");
    if (defer) {
      if ((rc = a.raw.alloc(vbytes))) return rc;
      if (vbytes && i < defer->values_dev.size() && defer->values_dev[i]) defer->copies.push_back({a.raw.p, defer->values_dev[i], vbytes});
      else if (vbytes) HIP_TRY(hipMemcpyAsync(a.raw.p, d.values, vbytes, hipMemcpyHostToDevice, defer->stream));
    } else if (stage_host) {
      void* dst = staged(a.raw, vbytes);
      if (!dst) return fail(DMI_ERR_HIP, "upload staging overflow");
      if (vbytes >= ((size_t)8 << 20)) parallel_for(vbytes, [&](size_t lo, size_t hi) { std::memcpy(static_cast<uint8_t*>(dst) + lo, static_cast<const uint8_t*>(d.values) + lo, hi - lo); });
      else if (vbytes) std::memcpy(dst, d.values, vbytes);
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
      if ((rc = a.aux_flags.alloc(((size_t)n / 64 + 4) * 4))) return rc;
      if (!a.aux_flags.pooled) HIP_TRY(hipMemsetAsync(a.aux_flags.p, 0, a.aux_flags.bytes, s));
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
    if ((rc = a.batch_flags.alloc(((size_t)a.n_sym / 64 + 4) * 4))) return rc;
    if (!a.batch_flags.pooled) HIP_TRY(hipMemsetAsync(a.batch_flags.p, 0, a.batch_flags.bytes, s));
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
  if (!job->slab.pooled) HIP_TRY(hipMemsetAsync(job->slab.p, 0, pinned_need, s));
  for (auto& a : job->atts) {
    uint8_t* base = job->slab.as<uint8_t>() + a.slab_off;
    a.small = SlabView{base, 64};
    a.meta = SlabView{base + 64, 64};
    a.hist = SlabView{base + 128, (size_t)a.bins_cap * 4};
    a.summary = SlabView{base + 128 + (size_t)a.bins_cap * 4, a.scheme == kTexCoord ? (size_t)std::max(1u, orient_summary_blocks(job->tables[a.table].n_seq)) * 16 : 0};
  }
  job->pinned_bytes = pinned_need;   // (the pinned mirror is allocated by the first single-job encode: a batch reads back through its arena)
  job->dev_tables = !std::getenv("DMI_HOST_TABLES");
  for (auto& a : job->atts) if (a.port == kToBits) job->dev_tables = false;
  {   // where the serial coders of a single-job encode run: DMI_CHAINS=device|host forces, default = by the longest stream
    // (a host core steps ≈ 6× faster than a scalar-unit walker, but costs a read-back of the symbols and a few thread starts)
    const char* m = std::getenv("DMI_CHAINS");
    uint64_t longest = 0;
    for (auto& a : job->atts) longest = std::max<uint64_t>(longest, a.n_sym);
    if (m && std::strcmp(m, "host") == 0) job->host_chains = true;
    else if (m && std::strcmp(m, "device") == 0) job->host_chains = false;
    else job->host_chains = longest >= kHostChainMinSymbols;
  }
  if ((rc = job->descs.alloc(sizeof(ChainDesc) * (size_t)n_atts * 2 + 16)   /* + the chain kernel's pull counter */)) return rc;
  if (cfg.flags & DMI_FLAG_TIMINGS) {
    for (auto& e : job->ev) HIP_TRY(hipEventCreate(&e));
    job->have_events = true;
  }
  uint32_t bad_p2v = 0;
  if (d_bad) HIP_TRY(hipMemcpyAsync(&bad_p2v, d_bad, 4, hipMemcpyDeviceToHost, s));
  if (!defer) HIP_TRY(hipStreamSynchronize(s));   // (a batch waits once, for all its jobs)
  if (bad_p2v) return fail(DMI_ERR_INVALID_ARGUMENT, "point_to_value entry out of range");
  if (trace_create) std::fprintf(stderr, "[dmi] job create (%u faces, %s relabelling): sequences %.1f ms, relabel + table uploads %.1f, attribute uploads + buffers + fan rows %.1f, stream + plan %.1f\n", F, device_relabel ? "device" : "host", t_seq, t_relabel, since_ms(tc0) - t_seq - t_relabel,
                                 std::chrono::duration<double, std::milli>(tc0 - t_enter).count());
  *job_out = job.release();
  return DMI_OK;
}

extern "C" {
void dmi_job_destroy(dmi_job* job) { delete job; }

int dmi_job_timings(const dmi_job* job, dmi_timings* t) {
  if (!job || !t) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  *t = job->last;
  return DMI_OK;
}

// The encode pipeline of one job, split at its host synchronisation points so that a batch of jobs can share
// them (one sync for all histograms, ONE k_chains launch holding every stream of every job).
static int check_value_bounds(const AttJob& a, const uint32_t* small, uint32_t i);
static int encode_phase_a(dmi_job* job, bool plan_only = false) {   // device: ranges → coding-order portabilization → predict → histograms; async read-back
  // plan_only: the caller has set a step sink — every launch below is collected, not issued, and the read-back is the caller's
  if (!plan_only) HIP_TRY(hipSetDevice(job->cfg.device));
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  if (!plan_only && !job->pinned) HIP_TRY(hipHostMalloc(&job->pinned, job->pinned_bytes, hipHostMallocDefault));
  uint8_t* pinned = static_cast<uint8_t*>(job->pinned);
  const bool timed = job->have_events;
  // ---- stage 1: value ranges (streamed over the unique values) ---------------------------------------------
  if (timed) HIP_TRY(hipEventRecord(job->ev[0], s));
  {
    RangeArgs ra{};
    for (auto& a : job->atts) {
      RangeAtt& r = ra.a[ra.count++];
      r.raw = a.raw.as<float>();
      r.partials = a.partials.as<float>();
      r.meta = a.meta.as<float>();
      r.small = a.small.as<uint32_t>();   // zeroed here; [0..1] := {INT_MAX, INT_MIN}; [4] := zero-length normal seen
      r.zero = a.meta.as<uint32_t>();   // meta, histogram, summaries: contiguous in the slab slot
      r.zero_words = (a.meta.bytes + a.hist.bytes + a.summary.bytes) / 4;
      r.n = a.desc.num_unique;
      r.N = a.desc.num_components;
      r.kind = a.port == kCoordwise ? 0 : (a.port == kOct ? 1 : 2);
      if (ra.count == kMaxRangeAtts) { launch_value_ranges(ra, s); ra.count = 0; }
    }
    launch_value_ranges(ra, s);
  }
  // ---- stage 2: portabilization in coding order + predict + transform ---------------------------------------
  if (timed) HIP_TRY(hipEventRecord(job->ev[1], s));
  // small: [0..1] minmax, [2..3] counters, [4] zero-normal flag, [5] hist overflow, [8..13] coder lengths / flags / ticks
  for (size_t ti = 0; ti < job->tables.size(); ++ti) {
    TableDev& t = job->tables[ti];
    if (t.alias_of >= 0) continue;
    QuantArgs qa{};
    auto flush = [&]() {
      if (qa.count) launch_seq_quantize(t.s2p.as<uint32_t>(), t.n_seq, qa, s);
      qa.count = 0;
    };
    for (auto& a : job->atts) {
      if ((size_t)a.table != ti) continue;
      QuantAtt& g = qa.a[qa.count++];
      g.raw = a.raw.as<float>();
      g.s2v = a.s2v.as<uint32_t>();
      g.qs = a.qs.p;
      g.fmt = a.qfmt;
      g.ipartials = a.ipartials.as<int32_t>();
      g.meta = a.meta.as<float>();
      g.maxq = (float)(uint64_t)((1ull << a.bits) - 1ull);
      g.kind = a.port == kCoordwise ? 0 : (a.port == kOct ? 1 : 2);
      g.N = a.desc.num_components;
      if (qa.count == kMaxGather) flush();
    }
    flush();
  }
  {
    MinMaxArgs ma{};
    for (auto& a : job->atts) {
      MinMaxAtt& m = ma.a[ma.count++];
      m.ipartials = a.ipartials.as<int32_t>();
      m.minmax = a.small.as<int32_t>();
      m.blocks = seq_quantize_blocks(job->tables[a.table].n_seq);
      if (ma.count == kMaxRangeAtts) { launch_i32_minmax_final(ma, s); ma.count = 0; }
    }
    launch_i32_minmax_final(ma, s);
  }
  OrientArgs fused_orient{};
  for (auto& a : job->atts) {
    const TableDev& t = job->tables[a.table];
    const int32_t* minmax = a.small.as<int32_t>();
    uint32_t* counters = a.small.as<uint32_t>() + 2;
    const uint32_t n = t.n_seq;
    if (n == 0) continue;
    if (a.fused_into >= 0) continue;   // predicted by its parent's fused sweep
    if (a.fused_nrm >= 0 || a.fused_uv >= 0) {
      FusedArgs fa{};
      fa.seq = t.seq.as<uint32_t>(); fa.c2r = t.c2r.as<uint32_t>(); fa.opp = t.opp.as<uint32_t>(); fa.n = n;
      fa.qs_pos = a.qs.p; fa.mm_pos = minmax; fa.sym_pos = a.sym.p;
      fa.packed = a.qfmt == QF_P64 ? 1u : 0u;
      fa.sym16 = a.sym16 ? 1u : 0u;
      fa.fan_hdr = t.fan_hdr.as<uint32_t>(); fa.fan_apex = t.fan_apex.as<uint32_t>(); fa.fan = t.fan.as<uint32_t>();
      if (a.fused_nrm >= 0) {
        AttJob& q = job->atts[a.fused_nrm];
        fa.qs_nrm = q.qs.p; fa.sym_nrm = q.sym.p; fa.flips = q.aux.as<uint8_t>(); fa.counters = q.small.as<uint32_t>() + 2;
        fa.flip_partials = q.flip_partials.as<uint32_t>(); q.flip_blocks = predict_fused_blocks(n);
        if (q.sym16) fa.sym16 |= 2u;
      }
      if (a.fused_uv >= 0) {
        AttJob& q = job->atts[a.fused_uv];
        fa.qs_uv = q.qs.p; fa.mm_uv = q.small.as<int32_t>(); fa.sym_uv = q.sym.p; fa.orient = q.aux.as<uint8_t>();
        fa.fix_list = q.fix_list.as<uint32_t>(); fa.fix_count = q.small.as<uint32_t>() + 3;
        if (q.sym16) fa.sym16 |= 4u;
      }
      launch_predict_fused(fa, s);
      if (a.fused_uv >= 0) { AttJob& q = job->atts[a.fused_uv]; fused_orient = OrientArgs{q.aux.as<uint8_t>(), q.summary.as<uint32_t>(), n, 0u}; }   // summarised by the histogram launch
      continue;
    }
    switch (a.scheme) {
      case kParallelogram:
        launch_pred_parallelogram_wrapped(t.seq.as<uint32_t>(), n, t.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), a.qs.as<int32_t>(), minmax, a.nq, a.sym.p, a.sym16, s);
        break;
      case kDelta:
        launch_pred_delta_difference(n, a.qs.as<int32_t>(), a.nq, a.sym.p, a.sym16, s);
        break;
      case kNormal: {
        // the fan-row sweep with only the normal attribute: fans of this (attribute) table, positions through the parent's table
        const AttJob& p = job->atts[a.parent];
        FusedArgs fa{};
        fa.seq = t.seq.as<uint32_t>(); fa.c2r = job->tables[p.table].c2r.as<uint32_t>(); fa.opp = t.opp.as<uint32_t>(); fa.n = n;
        fa.qs_pos = p.qs.p; fa.packed = p.qfmt == QF_P64 ? 1u : 0u;   // (the parent's positions may be packed by its own fused sweep; this attribute's values are not)
        fa.qs_nrm = a.qs.p; fa.sym_nrm = a.sym.p; fa.flips = a.aux.as<uint8_t>(); fa.counters = counters;
        fa.flip_partials = a.flip_partials.as<uint32_t>(); a.flip_blocks = predict_fused_blocks(n);
        fa.sym16 = a.sym16 ? 2u : 0u;
        fa.fan_hdr = a.fan_hdr.as<uint32_t>(); fa.fan_apex = a.fan_apex.as<uint32_t>(); fa.fan = a.fan.as<uint32_t>();
        launch_predict_fused(fa, s);
        break;
      }
      case kTexCoord: {
        const AttJob& p = job->atts[a.parent];
        launch_pred_texcoord_wrapped(t.seq.as<uint32_t>(), n, t.c2r.as<uint32_t>(), a.qs.as<int32_t>(), job->tables[p.table].c2r.as<uint32_t>(), p.qs.p, p.qfmt, minmax, a.sym.p, a.sym16, a.aux.as<uint8_t>(), s);
        launch_orient_summary(a.aux.as<uint8_t>(), n, a.summary.as<uint32_t>(), nullptr, s);
        break;
      }
    }
  }
  // ---- stage 3: histograms ---------------------------------------------------------------------------
  if (timed) HIP_TRY(hipEventRecord(job->ev[2], s));
  std::vector<size_t>& pin_off = job->run.pin_off;
  pin_off.assign(n_atts, 0);
  HistArgs ha{};
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    if (a.port == kToBits) {
      // alphabet bound needs the value range: read min/max first
      int32_t mm[2];
      HIP_TRY(hipMemcpyAsync(mm, a.small.p, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      const uint64_t span = (mm[1] >= mm[0]) ? (uint64_t)((int64_t)mm[1] - (int64_t)mm[0]) : 0;
      const uint64_t need = (a.transform == kWrapped ? span + 3 : 2 * span + 2);
      if (need > a.bins_cap) return fail(DMI_ERR_ALPHABET_TOO_LARGE, "custom attribute value range needs more than 2^20 symbols");
      a.bins = (uint32_t)need;
    }
    HistAtt& h = ha.a[ha.count++];
    h.sym = a.sym.p; h.sym16 = a.sym16 ? 1u : 0u; h.n = a.n_sym; h.hist = a.hist.as<uint32_t>(); h.bins = a.bins; h.overflow = a.small.as<uint32_t>() + 5;
    if (a.scheme == kNormal && a.flip_partials.p && a.n_sym) { h.flip_partials = a.flip_partials.as<uint32_t>(); h.flip_count = a.small.as<uint32_t>() + 2; h.n_flip_partials = a.flip_blocks; }
    if (ha.count == kMaxRangeAtts) { launch_histograms(ha, s); ha.count = 0; }
    pin_off[i] = a.slab_off;
  }
  ha.orient = fused_orient;
  launch_histograms(ha, s);
  // scratch words, ranges, histograms and orientation summaries of every attribute: one copy (the pinned buffer mirrors the slab)
  if (!plan_only && !job->dev_tables) HIP_TRY(hipMemcpyAsync(pinned, job->slab.p, job->slab.bytes, hipMemcpyDeviceToHost, s));
  if (timed) HIP_TRY(hipEventRecord(job->ev[3], s));
  return DMI_OK;
}

static int encode_phase_b(dmi_job* job, bool plan_only = false, bool host_chains = false) {   // host: table normalisation; device: coding records; fills job->run.descs
  // host_chains: the streams are coded on host cores (encode_tail_host): no coding records are built, orientation flags are compacted to bits
  // plan_only: a step sink is set — launches are collected, uploads are deferred to job->run.pending: no HIP call is made
  hipStream_t s = job->stream;
  job->run.pending.clear();
  auto upload_table = [&](void* dst, const void* src, size_t bytes) -> int {
    if (plan_only) { job->run.pending.push_back({dst, src, (bytes + 15) & ~(size_t)15}); return DMI_OK; }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    return DMI_OK;
  };
  const uint32_t n_atts = (uint32_t)job->atts.size();
  uint8_t* pinned = job->readback ? job->readback : static_cast<uint8_t*>(job->pinned);
  // ---- stage 4 (host): normalise tables, build chain descriptors -----------------------------------------
  std::vector<ChainDesc>& descs = job->run.descs;
  descs.clear();
  std::vector<AuxInfo>& aux = job->run.aux;
  aux.assign(n_atts, AuxInfo{});
  job->run.hdr_ptr.assign(n_atts, nullptr);
  job->run.hdr_len.assign(n_atts, 0);
  const std::vector<size_t>& pin_off = job->run.pin_off;
  std::string err;
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    const uint8_t* base = pinned + pin_off[i];
    const uint32_t* small = reinterpret_cast<const uint32_t*>(base);
    if (std::getenv("DMI_TRACE")) std::fprintf(stderr, "[dmi] attribute %u small: %08x %08x %u %u %u %u %u %u\n", i, small[0], small[1], small[2], small[3], small[4], small[5], small[6], small[7]);
    if (small[4]) return fail(DMI_ERR_ZERO_NORMAL, "attribute " + std::to_string(i) + " contains a zero-length normal (reference assert, geom.rs:45)");
    if (small[5]) return fail(DMI_ERR_ALPHABET_TOO_LARGE, "symbol outside the histogram bound");
    { const int brc = check_value_bounds(a, small, i); if (brc) return brc; }
    const uint32_t n = job->tables[a.table].n_seq;
    if (a.n_sym == 0) return fail(DMI_ERR_ENTROPY, "attribute " + std::to_string(i) + " has no values to code (empty histogram)");
    const uint32_t* hist = reinterpret_cast<const uint32_t*>(base + 128);
    int rc = a.ft.build(hist, a.bins, err);
    if (rc) return fail(rc, err);
    job->run.hdr_ptr[i] = a.ft.header.data();
    job->run.hdr_len[i] = (uint32_t)a.ft.header.size();
    std::vector<RansEntry>& rt = a.rt_host;
    rt.assign((a.ft.freq.size() + 3) & ~(size_t)3, RansEntry{0u, 0u, 0u, 0u, 0u});   // 4 entries = 80 bytes = whole 16-byte words
    for (size_t k = 0; k < a.ft.freq.size(); ++k) rt[k] = make_rans_entry(a.ft.freq[k], a.ft.cum[k], a.ft.precision);
    if (!host_chains) {
      { const int urc = upload_table(a.rtable.p, rt.data(), rt.size() * sizeof(RansEntry)); if (urc) return urc; }
      // symbols → coding records in coding order (data-parallel), consumed by the scalar chain
      launch_rans_prep(a.sym.p, a.sym16, a.n_sym, a.rtable.as<RansEntry>(), a.bins, a.rec.as<RansEntry>(), a.batch_flags.as<uint32_t>(), s);
    }
    ChainDesc d{};
    d.kind = 0; d.precision = a.ft.precision; d.n = a.n_sym; d.sym = a.sym.p; d.table = a.rec.as<RansEntry>(); d.state0 = 4u << a.ft.precision; d.batch_flags = a.batch_flags.as<uint32_t>();
    {   // which step the stream's walker uses: the one-byte step pays off when few batches of 64 hold a rare symbol (f < 2^(P-8))
      uint64_t rare = 0;
      for (size_t k = 0; k < a.ft.freq.size(); ++k)
        if (a.ft.freq[k] && ((uint64_t)a.ft.freq[k] << 8) < ((uint64_t)1 << a.ft.precision)) rare += hist[k];
      const double clean = std::pow(1.0 - (double)rare / (double)std::max<uint64_t>(a.n_sym, 1), 64.0);
      d.one_byte = clean > 0.8 ? 1u : 0u;
    }
    d.out = a.out.as<uint8_t>(); d.cap = a.out_cap; d.out_len = a.small.as<uint32_t>() + 8; d.ticks = a.small.as<uint32_t>() + 12;
    aux[i].rans_desc = (int)descs.size();
    descs.push_back(d);
    if (a.scheme == kNormal) {
      // mesh_normal_prediction.rs:147-150
      const uint32_t count_false = small[2];
      aux[i].zero_prob = zero_probability(count_false, (float)n);
      aux[i].count = n;
      ChainDesc r{};
      if (!host_chains) {   // rABS (rans.rs:91-108): bit 1 codes with f1 = 256 - p0 and offset 0, bit 0 with p0 and offset f1
        const uint32_t p0 = aux[i].zero_prob, f1 = 256u - p0;
        launch_bits_prep(a.aux.as<uint8_t>(), n, make_rans_entry(p0, f1, 8), make_rans_entry(f1, 0, 8), a.aux_rec.as<RansEntry>(), s);
        launch_batch_flags(a.aux_rec.as<RansEntry>(), n, nullptr, a.aux_flags.as<uint32_t>(), s);
      }
      r.kind = 1; r.n = n; r.precision = 8; r.state0 = 4096; r.table = a.aux_rec.as<RansEntry>(); r.force_generic = 0; r.batch_flags = a.aux_flags.as<uint32_t>(); r.out = a.aux_out.as<uint8_t>(); r.cap = a.aux_cap; r.out_len = a.small.as<uint32_t>() + 10; r.ticks = a.small.as<uint32_t>() + 13;
      aux[i].desc = (int)descs.size();
      descs.push_back(r);
    } else if (a.scheme == kTexCoord) {
      // stitch per-block summaries: len = Σ count, freq_count_0 = forward transitions with last := true
      const uint32_t nb = orient_summary_blocks(n);
      const uint32_t* sm = reinterpret_cast<const uint32_t*>(base + 128 + (size_t)a.bins_cap * 4);
      uint64_t len = 0, trans = 0;
      uint32_t last = 1;
      for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t cnt = sm[4 * b], first = sm[4 * b + 1], lastv = sm[4 * b + 2], tr = sm[4 * b + 3];
        if (!cnt) continue;
        if (first != last) ++trans;
        trans += tr;
        last = lastv;
        len += cnt;
      }
      aux[i].zero_prob = zero_probability(trans, (float)len + 0.001f);
      aux[i].count = (uint32_t)len;
      ChainDesc r{};
      {   // compact offsets + successor values per 4096-flag chunk, then flags → coding records on the device
        std::vector<uint32_t>& info = a.info_host;
        info.assign(2 * (size_t)std::max(1u, nb), 0u);
        uint32_t off = 0;
        for (uint32_t b = 0; b < nb; ++b) { info[2 * b] = off; off += sm[4 * b]; }
        uint32_t nextv = 1;   // `true` after the last valid entry
        for (uint32_t b = nb; b-- > 0;) { info[2 * b + 1] = nextv; if (sm[4 * b]) nextv = sm[4 * b + 1]; }
        info.resize((info.size() + 3) & ~(size_t)3, 0u);   // whole 16-byte words (the batch driver copies in uint4)
        { const int urc = upload_table(a.chunk_info.p, info.data(), info.size() * 4); if (urc) return urc; }
        const uint32_t p0 = aux[i].zero_prob, f1 = 256u - p0;
        if (host_chains) launch_orient_bits(a.aux.as<uint8_t>(), n, a.chunk_info.as<uint32_t>(), a.aux_bits.as<uint8_t>(), s);
        else {
          launch_orient_prep(a.aux.as<uint8_t>(), n, a.chunk_info.as<uint32_t>(), make_rans_entry(p0, f1, 8), make_rans_entry(f1, 0, 8), a.aux_rec.as<RansEntry>(), s);
          launch_batch_flags(a.aux_rec.as<RansEntry>(), len, nullptr, a.aux_flags.as<uint32_t>(), s);
        }
      }
      r.kind = 2; r.n = len; r.precision = 8; r.state0 = 4096; r.table = a.aux_rec.as<RansEntry>(); r.force_generic = 0; r.batch_flags = a.aux_flags.as<uint32_t>(); r.out = a.aux_out.as<uint8_t>(); r.cap = a.aux_cap; r.out_len = a.small.as<uint32_t>() + 10; r.ticks = a.small.as<uint32_t>() + 13;
      aux[i].desc = (int)descs.size();
      descs.push_back(r);
    }
  }
  return DMI_OK;
}

// Device form of phase B: one k_tables workgroup per attribute (normalisation, serialised table, coding records, metadata
// parameters, chain descriptors written to desc_base[…]), then the record prep — launches only, nothing waits for the host.
// run.descs keeps a host mirror of the static descriptor fields (stream lengths as the host knows them, capacities).
static uint32_t count_streams(const dmi_job* job) {
  uint32_t k = 0;
  for (const auto& a : job->atts) k += (a.scheme == kNormal || a.scheme == kTexCoord) ? 2u : 1u;
  return k;
}
static int encode_phase_b_dev(dmi_job* job, ChainDesc* desc_base, ChainDesc* hdr_desc_base, bool host_chains = false) {
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  std::vector<ChainDesc>& descs = job->run.descs;
  descs.clear();
  std::vector<AuxInfo>& aux = job->run.aux;
  aux.assign(n_atts, AuxInfo{});
  job->run.pending.clear();
  // a single job's table stage is ONE launch (a block per attribute); a batch collects per-attribute steps into its multi-item launch
  const bool grouped = !step_sink_active();
  std::unique_ptr<TableGroup> group(grouped ? new TableGroup() : nullptr);
  if (group) group->count = 0;
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    if (a.n_sym == 0) return fail(DMI_ERR_ENTROPY, "attribute " + std::to_string(i) + " has no values to code (empty histogram)");
  }
  for (int pass = 0; pass < 2; ++pass)
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    const uint32_t n = job->tables[a.table].n_seq;
    if (pass == 1) {   // what depends on the table kernel's outputs
      if (i == 0 && group) { launch_tables_group(*group, s); group->count = 0; }
      if (host_chains) {   // no coding records: the streams are coded on host cores from the symbols, the table and the metadata bits
        if (a.scheme == kTexCoord) launch_orient_bits(a.aux.as<uint8_t>(), n, a.chunk_info.as<uint32_t>(), a.aux_bits.as<uint8_t>(), s);
        continue;
      }
      launch_rans_prep(a.sym.p, a.sym16, a.n_sym, a.rtable.as<RansEntry>(), a.bins, a.rec.as<RansEntry>(), a.batch_flags.as<uint32_t>(), s);
      if (a.scheme == kNormal) {
        launch_bits_prep_dev(a.aux.as<uint8_t>(), n, a.aux_entries.as<RansEntry>(), a.aux_rec.as<RansEntry>(), s);
        launch_batch_flags(a.aux_rec.as<RansEntry>(), n, nullptr, a.aux_flags.as<uint32_t>(), s);
      } else if (a.scheme == kTexCoord) {
        launch_orient_prep_dev(a.aux.as<uint8_t>(), n, a.chunk_info.as<uint32_t>(), a.aux_entries.as<RansEntry>(), a.aux_rec.as<RansEntry>(), s);
        launch_batch_flags(a.aux_rec.as<RansEntry>(), n, a.small.as<uint32_t>() + 15, a.aux_flags.as<uint32_t>(), s);
      }
      continue;
    }
    TableAtt ta{};
    ta.hist = a.hist.as<uint32_t>(); ta.freq = a.freq.as<uint32_t>(); ta.rtable = a.rtable.as<RansEntry>(); ta.hdr = a.hdr.as<uint8_t>(); ta.small = a.small.as<uint32_t>();
    ta.n_sym = a.n_sym; ta.bins = a.bins; ta.hdr_cap = a.hdr_cap;
    aux[i].rans_desc = (int)descs.size();
    ta.desc = desc_base + descs.size();
    ta.sym = a.sym.p; ta.rec = a.rec.as<RansEntry>(); ta.batch_flags = a.batch_flags.as<uint32_t>(); ta.out = a.out.as<uint8_t>(); ta.out_cap = a.out_cap;
    ChainDesc d{};
    d.kind = 0; d.n = a.n_sym; d.out = a.out.as<uint8_t>(); d.cap = a.out_cap; d.out_len = a.small.as<uint32_t>() + 8;
    descs.push_back(d);
    if (a.scheme == kNormal || a.scheme == kTexCoord) {
      ta.aux_kind = a.scheme == kNormal ? 1u : 2u;
      ta.n_entries = n;
      ta.summary = a.summary.as<uint32_t>(); ta.chunk_info = a.chunk_info.as<uint32_t>(); ta.aux_entries = a.aux_entries.as<RansEntry>();
      ta.summary_blocks = a.scheme == kTexCoord ? orient_summary_blocks(n) : 0u;
      aux[i].desc = (int)descs.size();
      ta.aux_desc = desc_base + descs.size();
      ta.aux_rec = a.aux_rec.as<RansEntry>(); ta.aux_flags = a.aux_flags.as<uint32_t>(); ta.aux_out = a.aux_out.as<uint8_t>(); ta.aux_cap = a.aux_cap;
      ChainDesc r{};
      r.kind = ta.aux_kind; r.n = n; r.out = a.aux_out.as<uint8_t>(); r.cap = a.aux_cap; r.out_len = a.small.as<uint32_t>() + 10;
      descs.push_back(r);
    }
    ta.hdr_desc = hdr_desc_base ? hdr_desc_base + i : nullptr;
    if (group) {
      group->a[group->count++] = ta;
      if (group->count == kTableGroup) { launch_tables_group(*group, s); group->count = 0; }
    } else {
      launch_tables(ta, s);
    }
  }
  return DMI_OK;
}

// errors the device reports through an attribute's scratch words (device form; the host form meets them in phase B)
// The packed value layouts (QF_P64 / QF_H32) and the 16-bit symbols narrow what they store; what keeps that exact is the quantizer's bound
// 0 ≤ q < 2^bits (NaN → 0, ±inf → a range end).  The joint min/max every encode computes anyway (small[0..1]) is held to that bound here:
// a violated bound is an error, never a silently truncated field.
static int check_value_bounds(const AttJob& a, const uint32_t* small, uint32_t i) {
  if (a.port != kCoordwise || (a.qfmt == QF_I32 && !a.sym16)) return DMI_OK;
  const int32_t mn = (int32_t)small[0], mx = (int32_t)small[1];
  if (mn > mx) return DMI_OK;   // (no entries: the seeds)
  if (mn < 0 || (int64_t)mx > ((int64_t)1 << a.bits) - 1)
    return fail(DMI_ERR_ALPHABET_TOO_LARGE, "attribute " + std::to_string(i) + ": a quantized value lies outside [0, 2^bits) — packed layouts cannot hold it");
  return DMI_OK;
}
static int check_device_flags(const uint32_t* small, uint32_t i) {
  if (small[4]) return fail(DMI_ERR_ZERO_NORMAL, "attribute " + std::to_string(i) + " contains a zero-length normal (reference assert, geom.rs:45)");
  if (small[5]) return fail(DMI_ERR_ALPHABET_TOO_LARGE, "symbol outside the histogram bound");
  switch (small[7]) {
    case 0: return DMI_OK;
    case 1: return fail(DMI_ERR_ENTROPY, "empty symbol histogram");
    case 2: return fail(DMI_ERR_ENTROPY, "frequency normalisation overflow");
    case 3: return fail(DMI_ERR_ENTROPY, "frequency normalisation underflow");
    case 4: return fail(DMI_ERR_ENTROPY, "normalised frequency of an occurring symbol is zero (the reference encoder does not terminate on this input)");
    default: return fail(DMI_ERR_ENTROPY, "serialised frequency table exceeds its buffer");
  }
}

static int encode_phase_c1(dmi_job* job) {   // after the chains: async read-back of lengths / error flags
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  uint8_t* pinned = job->readback ? job->readback : static_cast<uint8_t*>(job->pinned);
  if (job->dev_tables) {   // nothing has come back yet: scratch words + quantization ranges of every attribute, 128 bytes each
    job->run.pin_off.assign(n_atts, 0);
    for (uint32_t i = 0; i < n_atts; ++i) {
      job->run.pin_off[i] = job->atts[i].slab_off;
      HIP_TRY(hipMemcpyAsync(pinned + job->run.pin_off[i], job->atts[i].small.p, 128, hipMemcpyDeviceToHost, s));
    }
    return DMI_OK;
  }
  for (uint32_t i = 0; i < n_atts; ++i) HIP_TRY(hipMemcpyAsync(pinned + job->run.pin_off[i], job->atts[i].small.p, 64, hipMemcpyDeviceToHost, s));
  return DMI_OK;
}

static int encode_phase_c2(dmi_job* job) {   // lengths known: async copy of the coded bytes
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  uint8_t* pinned = job->readback ? job->readback : static_cast<uint8_t*>(job->pinned);
  const std::vector<size_t>& pin_off = job->run.pin_off;
  const std::vector<AuxInfo>& aux = job->run.aux;
  auto& rans_off = job->run.rans_off;
  auto& aux_off = job->run.aux_off;
  rans_off.assign(n_atts, 0);
  aux_off.assign(n_atts, 0);
  job->run.rans_ptr.assign(n_atts, nullptr);
  job->run.aux_ptr.assign(n_atts, nullptr);
  job->run.rans_len.assign(n_atts, 0);
  job->run.aux_len.assign(n_atts, 0);
  size_t total = 0;
  std::vector<AuxInfo>& aux_w = job->run.aux;
  if (job->dev_tables) {
    job->run.hdr_ptr.assign(n_atts, nullptr);
    job->run.hdr_len.assign(n_atts, 0);
    job->run.hdr_off.assign(n_atts, 0);
    for (uint32_t i = 0; i < n_atts; ++i) {
      const uint32_t* small = reinterpret_cast<const uint32_t*>(pinned + pin_off[i]);
      int frc = check_device_flags(small, i);
      if (!frc) frc = check_value_bounds(job->atts[i], small, i);
      if (frc) return frc;
      aux_w[i].zero_prob = (uint8_t)small[14];
      aux_w[i].count = small[15];
      job->run.hdr_off[i] = total; total += (small[6] + 15u) & ~15u;
    }
  }
  for (uint32_t i = 0; i < n_atts; ++i) {
    const uint32_t* small = reinterpret_cast<const uint32_t*>(pinned + pin_off[i]);
    if (small[9] || small[11]) return fail(DMI_ERR_ENTROPY, small[9] == 1 || small[11] == 1 ? "rANS state too large" : "coder output capacity exceeded");
    if (std::getenv("DMI_TRACE")) std::fprintf(stderr, "[dmi] attribute %u: rANS chain %.3f ms (%llu symbols), aux chain %.3f ms\n", i, small[12] * 1e-5, (unsigned long long)job->atts[i].n_sym, small[13] * 1e-5);
    rans_off[i] = total; total += (small[8] + 15u) & ~15u;
    if (aux[i].desc >= 0) { aux_off[i] = total; total += (small[10] + 15u) & ~15u; }
  }
  if (total > job->out_pinned_cap) {
    if (job->out_pinned) (void)hipHostFree(job->out_pinned);
    job->out_pinned = nullptr;
    job->out_pinned_cap = total + total / 4 + 4096;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&job->out_pinned), job->out_pinned_cap, hipHostMallocDefault));
  }
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    const uint32_t* small = reinterpret_cast<const uint32_t*>(pinned + pin_off[i]);
    if (small[8]) HIP_TRY(hipMemcpyAsync(job->out_pinned + rans_off[i], a.out.p, small[8], hipMemcpyDeviceToHost, s));
    if (aux[i].desc >= 0 && small[10]) HIP_TRY(hipMemcpyAsync(job->out_pinned + aux_off[i], a.aux_out.p, small[10], hipMemcpyDeviceToHost, s));
    job->run.rans_ptr[i] = job->out_pinned + rans_off[i];
    job->run.aux_ptr[i] = job->out_pinned + aux_off[i];
    job->run.rans_len[i] = small[8];
    job->run.aux_len[i] = aux[i].desc >= 0 ? small[10] : 0u;
    if (job->dev_tables) {
      if (small[6]) HIP_TRY(hipMemcpyAsync(job->out_pinned + job->run.hdr_off[i], a.hdr.p, small[6], hipMemcpyDeviceToHost, s));
      job->run.hdr_ptr[i] = job->out_pinned + job->run.hdr_off[i];
      job->run.hdr_len[i] = small[6];
    }
  }
  return DMI_OK;
}

// Batch form of c1 + c2: lengths, error flags and bytes come from the packed arena (one table + one byte copy per batch).
static int encode_phase_c_packed(dmi_job* job, const PackEntry* table, uint32_t first_desc, const uint8_t* arena_host) {
  const uint32_t n_atts = (uint32_t)job->atts.size();
  const std::vector<AuxInfo>& aux = job->run.aux;
  job->run.rans_ptr.assign(n_atts, nullptr);
  job->run.aux_ptr.assign(n_atts, nullptr);
  job->run.rans_len.assign(n_atts, 0);
  job->run.aux_len.assign(n_atts, 0);
  for (uint32_t i = 0; i < n_atts; ++i) {
    const PackEntry& r = table[first_desc + (uint32_t)aux[i].rans_desc];
    if (r.err) return fail(DMI_ERR_ENTROPY, r.err == 1 ? "rANS state too large" : "coder output capacity exceeded");
    job->run.rans_ptr[i] = arena_host + r.offset;
    job->run.rans_len[i] = r.len;
    if (aux[i].desc >= 0) {
      const PackEntry& x = table[first_desc + (uint32_t)aux[i].desc];
      if (x.err) return fail(DMI_ERR_ENTROPY, x.err == 1 ? "rABS state too large" : "coder output capacity exceeded");
      job->run.aux_ptr[i] = arena_host + x.offset;
      job->run.aux_len[i] = x.len;
    }
  }
  return DMI_OK;
}

static int encode_phase_c3(dmi_job* job, dmi_buffer* out) {   // host: splice the attribute section
  const uint32_t n_atts = (uint32_t)job->atts.size();
  uint8_t* pinned = job->readback ? job->readback : static_cast<uint8_t*>(job->pinned);
  const std::vector<size_t>& pin_off = job->run.pin_off;
  const std::vector<AuxInfo>& aux = job->run.aux;
  const auto& rans_ptr = job->run.rans_ptr;
  const auto& aux_ptr = job->run.aux_ptr;
  // ---- stage 6 (host): splice the attribute section (encode/attribute/mod.rs:26-57, attribute_encoder.rs:159-160,344-386)
  // written straight into the caller's buffer: its size is bounded by the parts (< 96 bytes of framing per attribute)
  size_t bound = 16;
  for (uint32_t i = 0; i < n_atts; ++i) bound += 96 + (size_t)job->run.hdr_len[i] + job->run.rans_len[i] + job->run.aux_len[i];
  struct RawSink {
    uint8_t* p; size_t n = 0;
    void u8(uint8_t v) { p[n++] = v; }
    void u32(uint32_t v) { std::memcpy(p + n, &v, 4); n += 4; }   // little-endian host
    void f32(float f) { std::memcpy(p + n, &f, 4); n += 4; }
    void leb128(uint64_t v) { do { uint8_t x = v & 0x7F; v >>= 7; u8(v ? (x | 0x80) : x); } while (v); }
    void bytes(const uint8_t* q, size_t k) {
      if (k >= ((size_t)32 << 20)) {   // a large stream (≈ 100M-triangle meshes): the copy — and the first touch of the output pages — on a few threads
        const size_t parts = std::min<size_t>(8, k >> 22);
        std::vector<std::thread> th;
        for (size_t t = 0; t < parts; ++t) th.emplace_back([=] { const size_t lo = k * t / parts, hi = k * (t + 1) / parts; std::memcpy(p + n + lo, q + lo, hi - lo); });
        for (auto& x : th) x.join();
      } else if (k) {
        std::memcpy(p + n, q, k);
      }
      n += k;
    }
    void bytes(const std::vector<uint8_t>& v) { bytes(v.data(), v.size()); }
  } w{static_cast<uint8_t*>(std::malloc(bound))};
  if (!w.p) return fail(DMI_ERR_OUT_OF_MEMORY, "malloc");
  job->last_fixups = 0;
  for (uint32_t i = 0; i < n_atts; ++i)
    if (job->atts[i].scheme == kTexCoord && job->atts[i].fused_into >= 0) job->last_fixups += reinterpret_cast<const uint32_t*>(pinned + pin_off[i])[3];
  w.u8((uint8_t)n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) { w.u8((uint8_t)((uint8_t)i - 1)); w.u8(job->atts[i].desc.domain); w.u8(0); }   // Q13
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = job->atts[i];
    w.u8(1); w.u8(a.desc.att_type); w.u8(a.desc.component_type); w.u8(a.desc.num_components); w.u8(0); w.u8((uint8_t)a.desc.unique_id); w.u8((uint8_t)a.port);
  }
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = job->atts[i];
    const uint8_t* base = pinned + pin_off[i];
    // (pinned[off..off+64) holds `small` as read back after phase A — or again after the chains: min/max sit at the same offsets)
    const int32_t* mm = reinterpret_cast<const int32_t*>(base);
    const float* meta = reinterpret_cast<const float*>(base + 64);
    w.u8((uint8_t)a.scheme);
    w.u8((uint8_t)a.transform);
    w.u8(1);   // rans_encoding
    w.bytes(job->run.hdr_ptr[i], job->run.hdr_len[i]);
    const uint32_t rans_len = job->run.rans_len[i], aux_len = job->run.aux_len[i];
    w.leb128(rans_len);
    w.bytes(rans_ptr[i], rans_len);
    ByteSink tinfo;
    if (a.transform == kWrapped) { tinfo.u32((uint32_t)mm[0]); tinfo.u32((uint32_t)mm[1]); }
    else if (a.transform == kOctOrth) { tinfo.u32(255); tinfo.u32(127); }
    if (a.scheme == kNormal) {
      w.bytes(tinfo.b);
      w.u8(aux[i].zero_prob);
      w.leb128(aux_len);
      w.bytes(aux_ptr[i], aux_len);
    } else if (a.scheme == kTexCoord) {
      w.u32(aux[i].count);
      w.u8(aux[i].zero_prob);
      w.leb128(aux_len);
      w.bytes(aux_ptr[i], aux_len);
      w.bytes(tinfo.b);
    } else {
      w.bytes(tinfo.b);
    }
    if (a.port == kCoordwise) {   // quantization_coordinate_wise.rs:56-59
      for (int k = 0; k < a.desc.num_components; ++k) w.f32(meta[k]);
      w.f32(meta[a.desc.num_components]);
      w.u8((uint8_t)a.bits);
    } else if (a.port == kOct) {
      w.u8(8);                     // octahedral_quantization.rs:43
    }
  }
  out->data = w.p;
  out->len = w.n;
  out->cap = bound;
  return DMI_OK;
}

// Hybrid tail of a single-job encode (job->host_chains): after the table stage the symbols, the device-built coding tables, the
// serialised tables and the metadata bits come back into pinned staging (largest attribute first, one event per attribute) and every
// stream is coded by host_rans_chain / host_rabs_chain on its own host core as soon as its attribute has arrived; then the splice.
// The strict dependency chain of one stream is the only stage that leaves the device: a 15M-symbol stream takes ≈ 240 ms on a
// scalar-unit walker and ≈ 40 ms on one 5 GHz core.  Batches (dmi_jobs_encode) keep the device chains.
static int encode_tail_host(dmi_job* job, dmi_buffer* out, float* chain_ms, float* longest_ms, float* wait_ms) {
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  const bool dev = job->dev_tables;
  struct Slot { size_t small = 0, sym = 0, table = 0, hdr = 0, bits = 0; };
  std::vector<Slot> slot(n_atts);
  size_t need = 0;
  auto take = [&](size_t bytes) { const size_t at = need; need = (need + bytes + 255) & ~(size_t)255; return at; };
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = job->atts[i];
    const bool has_aux = a.scheme == kNormal || a.scheme == kTexCoord;
    slot[i].small = take(128);
    slot[i].sym = take((size_t)a.n_sym * (a.sym16 ? 2 : 4));
    if (dev) { slot[i].table = take((size_t)a.bins * sizeof(RansEntry)); slot[i].hdr = take(a.hdr_cap); }
    if (has_aux) slot[i].bits = take((size_t)job->tables[a.table].n_seq + 16);
  }
  if (!job->stage || job->stage->cap < need) {
    release_stage(job->stage);
    job->stage = acquire_stage(job->cfg.device, need);
    if (!job->stage) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (host-chain staging)");
  }
  uint8_t* base = job->stage->p;
  while (job->copy_ev.size() < n_atts) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); job->copy_ev.push_back(e); }
  while (job->host_out.size() < 2 * (size_t)n_atts) job->host_out.emplace_back(new HostChainOut());
  std::vector<uint32_t> order(n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return job->atts[x].n_sym > job->atts[y].n_sym; });
  for (uint32_t i : order) {
    AttJob& a = job->atts[i];
    const uint32_t n = job->tables[a.table].n_seq;
    if (dev) {
      HIP_TRY(hipMemcpyAsync(base + slot[i].small, a.small.p, 128, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(base + slot[i].table, a.rtable.p, (size_t)a.bins * sizeof(RansEntry), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(base + slot[i].hdr, a.hdr.p, a.hdr_cap, hipMemcpyDeviceToHost, s));
    }
    if (a.n_sym) HIP_TRY(hipMemcpyAsync(base + slot[i].sym, a.sym.p, (size_t)a.n_sym * (a.sym16 ? 2 : 4), hipMemcpyDeviceToHost, s));
    if (a.scheme == kNormal && n) HIP_TRY(hipMemcpyAsync(base + slot[i].bits, a.aux.p, n, hipMemcpyDeviceToHost, s));
    if (a.scheme == kTexCoord && n) HIP_TRY(hipMemcpyAsync(base + slot[i].bits, a.aux_bits.p, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(job->copy_ev[i], s));
  }
  // streams, longest first; a few host threads pull them
  struct Stream { uint32_t att; bool aux; uint64_t n; };
  std::vector<Stream> streams;
  for (uint32_t i : order) {
    const AttJob& a = job->atts[i];
    streams.push_back({i, false, a.n_sym});
    if (a.scheme == kNormal || a.scheme == kTexCoord) streams.push_back({i, true, job->tables[a.table].n_seq});
  }
  std::stable_sort(streams.begin(), streams.end(), [](const Stream& x, const Stream& y) { return x.n > y.n; });
  std::vector<int> rcs(streams.size(), DMI_OK);
  std::vector<std::string> errs(streams.size());
  std::vector<AuxInfo>& aux = job->run.aux;
  const int device = job->cfg.device;
  std::atomic<size_t> next{0};
  const auto t_chain0 = std::chrono::steady_clock::now();
  auto work = [&] {
    (void)hipSetDevice(device);
    for (size_t k; (k = next.fetch_add(1)) < streams.size();) {
      const Stream& st = streams[k];
      const uint32_t i = st.att;
      AttJob& a = job->atts[i];
      const auto w0 = std::chrono::steady_clock::now();
      if (hipEventSynchronize(job->copy_ev[i]) != hipSuccess) { rcs[k] = DMI_ERR_HIP; errs[k] = "hipEventSynchronize (host-chain staging)"; continue; }
      const auto w1 = std::chrono::steady_clock::now();
      const uint32_t* small = reinterpret_cast<const uint32_t*>(base + slot[i].small);
      if (dev) {
        int frc = check_device_flags(small, i);
        if (!frc) frc = check_value_bounds(a, small, i);
        if (frc) { rcs[k] = frc; errs[k] = g_last_error; continue; }
      }
      HostChainOut& o = *job->host_out[2 * (size_t)i + (st.aux ? 1 : 0)];
      if (!st.aux) {
        const RansEntry* table = dev ? reinterpret_cast<const RansEntry*>(base + slot[i].table) : a.rt_host.data();
        const uint32_t bins = dev ? a.bins : (uint32_t)a.ft.freq.size();
        const uint32_t precision = dev ? small[12] : a.ft.precision;
        if (a.sym16) host_rans_chain16(reinterpret_cast<const uint16_t*>(base + slot[i].sym), a.n_sym, table, bins, precision, o);
        else host_rans_chain(reinterpret_cast<const uint32_t*>(base + slot[i].sym), a.n_sym, table, bins, precision, o);
      } else {
        const uint32_t p0 = dev ? small[14] : aux[i].zero_prob, f1 = 256u - p0;
        const uint64_t count = dev ? small[15] : aux[i].count;
        const RansEntry e[2] = {make_rans_entry(p0, f1, 8), make_rans_entry(f1, 0, 8)};   // rABS (rans.rs:91-108): bit 0 codes with p0 and offset f1, bit 1 with f1 and offset 0
        host_rabs_chain(base + slot[i].bits, count, e, o);
      }
      if (k == 0) {   // the longest stream
        if (wait_ms) *wait_ms = std::chrono::duration<float, std::milli>(w1 - w0).count();
        if (longest_ms) *longest_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w1).count();
      }
      if (o.err) {
        rcs[k] = o.err == 2 ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_ENTROPY;
        errs[k] = o.err == 1 ? (st.aux ? "rABS state too large" : "rANS state too large") : (o.err == 2 ? "malloc (host-chain output)" : "symbol outside the coding table");
      }
    }
  };
  {
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({streams.size(), (size_t)host_threads(), (size_t)16}));
    std::vector<std::thread> th;
    for (size_t t = 1; t < n_threads; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (chain_ms) *chain_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_chain0).count();
  for (size_t k = 0; k < streams.size(); ++k) if (rcs[k]) return fail(rcs[k], errs[k]);
  // hand the parts to the splice
  job->run.rans_ptr.assign(n_atts, nullptr); job->run.aux_ptr.assign(n_atts, nullptr);
  job->run.rans_len.assign(n_atts, 0); job->run.aux_len.assign(n_atts, 0);
  if (dev) {
    job->readback = base;
    job->run.pin_off.assign(n_atts, 0);
    job->run.hdr_ptr.assign(n_atts, nullptr);
    job->run.hdr_len.assign(n_atts, 0);
  }
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = job->atts[i];
    const HostChainOut& r = *job->host_out[2 * (size_t)i];
    if (r.len > 0xFFFFFFFFull) return fail(DMI_ERR_ENTROPY, "coded stream exceeds 4 GiB");
    job->run.rans_ptr[i] = r.data; job->run.rans_len[i] = (uint32_t)r.len;
    if (a.scheme == kNormal || a.scheme == kTexCoord) {
      const HostChainOut& x = *job->host_out[2 * (size_t)i + 1];
      job->run.aux_ptr[i] = x.data; job->run.aux_len[i] = (uint32_t)x.len;
    }
    if (dev) {
      const uint32_t* small = reinterpret_cast<const uint32_t*>(base + slot[i].small);
      job->run.pin_off[i] = slot[i].small;
      job->run.hdr_ptr[i] = base + slot[i].hdr;
      job->run.hdr_len[i] = small[6];
      aux[i].zero_prob = (uint8_t)small[14];
      aux[i].count = small[15];
    }
  }
  return encode_phase_c3(job, out);
}

// Phase A as one hipGraph replay (single-job re-encodes and the jobs of a batch that keep their own launches): the ≈9 launches and
// the read-back of a job collapse into a single API call.  Jobs with
// event timing or a ToBits attribute (whose alphabet bound needs a mid-phase host wait) stay on the eager path.
static int run_phase_a(dmi_job* job) {
  hipStream_t s = job->stream;
  if (!job->pinned) { HIP_TRY(hipSetDevice(job->cfg.device)); HIP_TRY(hipHostMalloc(&job->pinned, job->pinned_bytes, hipHostMallocDefault)); }   // (not inside a stream capture)
  if (job->graph_a) { HIP_TRY(hipSetDevice(job->cfg.device)); HIP_TRY(hipGraphLaunch(job->graph_a, s)); return DMI_OK; }
  bool eligible = !job->have_events && !job->graph_tried;
  for (auto& a : job->atts) if (a.port == kToBits) eligible = false;
  if (!eligible) return encode_phase_a(job);
  job->graph_tried = true;
  HIP_TRY(hipSetDevice(job->cfg.device));
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return encode_phase_a(job);
  const int rc = encode_phase_a(job);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess || !graph) return encode_phase_a(job);
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) return encode_phase_a(job);
  job->graph_a = exec;
  HIP_TRY(hipGraphLaunch(exec, s));
  return DMI_OK;
}

int dmi_job_encode(dmi_job* job, dmi_buffer* out) {
  if (!job || !out) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  hipStream_t s = job->stream;
  const bool timed = job->have_events;
  const auto wall0 = std::chrono::steady_clock::now();
  job->readback = nullptr;
  // (a failure after the first launch waits for the stream before it returns: the caller may destroy the job at once)
  struct Drain { hipStream_t s; bool armed = true; ~Drain() { if (armed) (void)hipStreamSynchronize(s); } } drain{s};
  int rc = encode_phase_a(job);
  if (rc) return rc;
  auto t_tab0 = std::chrono::steady_clock::now(), t_tab1 = t_tab0;
  const bool host_chains = job->host_chains;
  if (job->dev_tables) {
    // tables, metadata parameters and descriptors on the device: the stream runs from the first kernel to the chains without a host wait
    if ((rc = encode_phase_b_dev(job, job->descs.as<ChainDesc>(), nullptr, host_chains))) return rc;
  } else {
    HIP_TRY(hipStreamSynchronize(s));
    t_tab0 = std::chrono::steady_clock::now();
    if ((rc = encode_phase_b(job, false, host_chains))) return rc;
    if (!host_chains) HIP_TRY(hipMemcpyAsync(job->descs.p, job->run.descs.data(), job->run.descs.size() * sizeof(ChainDesc), hipMemcpyHostToDevice, s));
    t_tab1 = std::chrono::steady_clock::now();
  }
  const std::vector<ChainDesc>& descs = job->run.descs;
  if (timed) HIP_TRY(hipEventRecord(job->ev[4], s));
  float host_chain_ms = 0.0f, longest_ms = 0.0f, wait_ms = 0.0f;
  if (host_chains) {
    // hybrid form: symbols + tables back over PCIe, every stream on a host core, splice
    if ((rc = encode_tail_host(job, out, &host_chain_ms, &longest_ms, &wait_ms))) { (void)hipStreamSynchronize(s); return rc; }
  } else {
    uint64_t longest = 0, total = 0;
    for (const ChainDesc& cd : descs) { longest = std::max<uint64_t>(longest, cd.n); total += cd.n; }
    launch_chains(job->descs.as<ChainDesc>(), nullptr, (uint32_t)descs.size(), reinterpret_cast<uint32_t*>(job->descs.as<ChainDesc>() + job->atts.size() * 2),
                  chain_launch_sparse(longest, total, (uint32_t)descs.size()), s);
  }
  if (timed) HIP_TRY(hipEventRecord(job->ev[5], s));
  if (!host_chains) {
    if ((rc = encode_phase_c1(job))) { (void)hipStreamSynchronize(s); return rc; }
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = encode_phase_c2(job))) { (void)hipStreamSynchronize(s); return rc; }
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = encode_phase_c3(job, out))) return rc;
  }

  dmi_timings tm{};
  if (timed) {
    HIP_TRY(hipEventSynchronize(job->ev[5]));
    (void)hipEventElapsedTime(&tm.quantize_ms, job->ev[0], job->ev[1]);
    (void)hipEventElapsedTime(&tm.predict_ms, job->ev[1], job->ev[2]);
    (void)hipEventElapsedTime(&tm.histogram_ms, job->ev[2], job->ev[3]);
    (void)hipEventElapsedTime(&tm.rans_ms, job->ev[4], job->ev[5]);
  }
  if (host_chains) tm.rans_ms = host_chain_ms;   // read-back of symbols / tables + the host-core chains (wall clock)
  tm.table_ms = std::chrono::duration<float, std::milli>(t_tab1 - t_tab0).count();
  if (timed && job->dev_tables) (void)hipEventElapsedTime(&tm.table_ms, job->ev[3], job->ev[4]);   // k_tables + record prep on the device
  tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  tm.predict_bytes = job->predict_bytes;
  for (auto& a : job->atts) tm.symbols += a.n_sym;
  tm.num_streams = host_chains ? count_streams(job) : (uint32_t)descs.size();
  tm.host_chains = host_chains ? 1u : 0u;
  tm.longest_stream_ms = longest_ms;
  tm.readback_wait_ms = wait_ms;
  tm.texcoord_fixups = job->last_fixups;
  job->last = tm;
  drain.armed = false;   // every path above ended with a stream synchronisation
  return DMI_OK;
}

// Batch form: every job's data-parallel stages are queued back to back, the host waits once, and all rANS/rABS
// streams of all jobs run in ONE k_chains launch (thousands of wavefronts — the regime where the one-wavefront-
// per-stream coder fills the chip).  All jobs must live on the same device; they are serialised on jobs[0]'s
// stream order-wise by using each job's own stream only when they are the same stream (see dmi_encode_attributes_batch).
// Device + pinned staging of one batch read-back, kept between calls (grow-only; a small pool so that concurrent
// dmi_jobs_encode calls do not share one).
struct BatchArena {
  int device = -1;
  void* bytes_dev = nullptr; size_t bytes_dev_cap = 0;
  void* table_dev = nullptr; void* table_host = nullptr; size_t table_cap = 0;
  void* bytes_host = nullptr; size_t bytes_host_cap = 0;
  void* plan_host = nullptr; void* plan_dev = nullptr; size_t plan_cap = 0;       // launch plan of a batch (argument blocks, block maps)
  void* slabs_host = nullptr; void* slabs_dev = nullptr; size_t slabs_cap = 0;   // every job's slab, packed
  void* descs_dev = nullptr; size_t descs_cap = 0;   // device form: chain descriptors | header pseudo-descriptors | stream order | pull counter
  int reserve_descs(size_t bytes) {
    if (bytes <= descs_cap) return DMI_OK;
    if (descs_dev) (void)hipFree(descs_dev);
    descs_dev = nullptr; descs_cap = 0;
    HIP_TRY(hipMalloc(&descs_dev, bytes + bytes / 4 + 4096));
    descs_cap = bytes + bytes / 4 + 4096;
    return DMI_OK;
  }
  bool in_use = false;
  int reserve(size_t dev_bytes, size_t table_bytes) {
    if (dev_bytes > bytes_dev_cap) {
      if (bytes_dev) (void)hipFree(bytes_dev);
      bytes_dev = nullptr; bytes_dev_cap = 0;
      HIP_TRY(hipMalloc(&bytes_dev, dev_bytes + dev_bytes / 4 + 4096));
      bytes_dev_cap = dev_bytes + dev_bytes / 4 + 4096;
    }
    if (table_bytes > table_cap) {
      if (table_dev) (void)hipFree(table_dev);
      if (table_host) (void)hipHostFree(table_host);
      table_dev = table_host = nullptr; table_cap = 0;
      HIP_TRY(hipMalloc(&table_dev, table_bytes * 2));
      HIP_TRY(hipHostMalloc(&table_host, table_bytes * 2, hipHostMallocDefault));
      table_cap = table_bytes * 2;
    }
    return DMI_OK;
  }
  static int reserve_pair(void*& host, void*& dev, size_t& cap, size_t bytes) {
    if (bytes <= cap) return DMI_OK;
    if (host) (void)hipHostFree(host);
    if (dev) (void)hipFree(dev);
    host = dev = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipHostMalloc(&host, want, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&dev, want));
    cap = want;
    return DMI_OK;
  }
  int reserve_host(size_t bytes) {
    if (bytes > bytes_host_cap) {
      if (bytes_host) (void)hipHostFree(bytes_host);
      bytes_host = nullptr; bytes_host_cap = 0;
      HIP_TRY(hipHostMalloc(&bytes_host, bytes + bytes / 4 + 4096, hipHostMallocDefault));
      bytes_host_cap = bytes + bytes / 4 + 4096;
    }
    return DMI_OK;
  }
};
static std::mutex g_arena_mutex;
static std::vector<BatchArena*> g_arenas;   // (never freed: process-lifetime staging)
static BatchArena* acquire_batch_arena(int device) {
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  for (BatchArena* a : g_arenas) if (!a->in_use && a->device == device) { a->in_use = true; return a; }
  BatchArena* a = new BatchArena();
  a->device = device;
  a->in_use = true;
  g_arenas.push_back(a);
  return a;
}
static void release_batch_arena(BatchArena* a) {
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  a->in_use = false;
}

// A batch's launch plan: the KernelSteps of many jobs grouped by (level, kernel); argument blocks, block maps and any extra
// tables go to the device in ONE copy, then every group is one multi-item launch.
struct BatchPlan {
  struct Group { int level = 0, id = 0; uint32_t lds = 0, total_blocks = 0; std::vector<const KernelStep*> items; size_t off_args = 0, off_info = 0, off_blocks = 0; };
  std::vector<Group> groups;
  size_t bytes = 0;
  static size_t align(size_t v) { return (v + 255) & ~(size_t)255; }
  void add(const std::vector<std::vector<KernelStep>>& steps, int n_levels) {
    // one pass: bucket (level, kernel) → group, in job order; then the groups in (level, kernel) order
    std::vector<Group> bucket((size_t)n_levels * K_COUNT);
    for (const auto& job_steps : steps)
      for (const KernelStep& st : job_steps) {
        if (st.level < 0 || st.level >= n_levels || st.id < 0 || st.id >= K_COUNT) continue;
        Group& g = bucket[(size_t)st.level * K_COUNT + st.id];
        g.items.push_back(&st); g.total_blocks += st.blocks; g.lds = std::max(g.lds, st.lds);
      }
    for (int level = 0; level < n_levels; ++level)
      for (int id = 0; id < K_COUNT; ++id) {
        Group& g = bucket[(size_t)level * K_COUNT + id];
        if (g.items.empty()) continue;
        g.level = level; g.id = id;
        groups.push_back(std::move(g));
      }
    for (Group& g : groups) {
      g.off_args = bytes; bytes = align(bytes + (size_t)g.items.size() * g.items[0]->args_size);
      g.off_info = bytes; bytes = align(bytes + (size_t)g.total_blocks * sizeof(uint2));
      g.off_blocks = bytes; bytes = align(bytes + g.items.size() * sizeof(uint32_t));
    }
  }
  size_t reserve(size_t n) { const size_t off = bytes; bytes = align(bytes + n); return off; }
  void fill_group(uint8_t* ph, const Group& g) const {
    const size_t asz = g.items[0]->args_size;
    uint2* info = reinterpret_cast<uint2*>(ph + g.off_info);
    uint32_t* blocks = reinterpret_cast<uint32_t*>(ph + g.off_blocks);
    uint32_t at = 0;
    for (size_t i = 0; i < g.items.size(); ++i) {
      std::memcpy(ph + g.off_args + i * asz, g.items[i]->args, asz);
      blocks[i] = g.items[i]->blocks;
      for (uint32_t b = 0; b < g.items[i]->blocks; ++b) info[at++] = make_uint2((uint32_t)i, b);
    }
  }
  void fill(uint8_t* ph) const {
    if (groups.size() < 4) { for (const Group& g : groups) fill_group(ph, g); return; }
    std::atomic<size_t> next{0};   // groups differ a lot in size: a few host threads pull them
    auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < groups.size();) fill_group(ph, groups[k]); };
    std::vector<std::thread> th;
    for (size_t t = 1; t < std::min<size_t>(groups.size(), 8); ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
  }
  void launch(const uint8_t* pd, hipStream_t s) const {
    for (const Group& g : groups)
      launch_steps_multi(g.id, pd + g.off_args, reinterpret_cast<const uint2*>(pd + g.off_info), reinterpret_cast<const uint32_t*>(pd + g.off_blocks), g.total_blocks, g.lds, s);
  }
};

// Phase A of a whole batch in ONE launch per (level, kernel): every job's launches are collected as KernelSteps (the same
// code path as a single encode, with a sink set), grouped, uploaded in one copy and served by multi-item kernels; the slabs
// come back packed in one copy.  Small meshes are otherwise bound by the ≈2.4 µs the GPU spends per tiny kernel (9 per job).
static int run_phase_a_batched(dmi_job** jobs, const std::vector<uint32_t>& which, BatchArena* arena, hipStream_t s) {
  const uint32_t n = (uint32_t)which.size();
  if (!n) return DMI_OK;
  std::vector<std::vector<KernelStep>> steps(n);
  for (uint32_t k = 0; k < n; ++k) {
    dmi_job* job = jobs[which[k]];
    set_step_sink(&steps[k]);
    const int rc = encode_phase_a(job, true);
    set_step_sink(nullptr);
    if (rc) return rc;
  }
  BatchPlan plan;
  plan.add(steps, kStepLevels);
  std::vector<CopyItem> copies(n);
  size_t slab_bytes = 0;
  for (uint32_t k = 0; k < n; ++k) {
    dmi_job* job = jobs[which[k]];
    copies[k] = CopyItem{job->slab.p, (uint64_t)slab_bytes, (uint64_t)(job->slab.bytes & ~(size_t)15)};
    slab_bytes = BatchPlan::align(slab_bytes + job->slab.bytes);
  }
  const size_t off_copies = plan.reserve(copies.size() * sizeof(CopyItem));
  int rc;
  if ((rc = BatchArena::reserve_pair(arena->plan_host, arena->plan_dev, arena->plan_cap, plan.bytes))) return rc;
  if ((rc = BatchArena::reserve_pair(arena->slabs_host, arena->slabs_dev, arena->slabs_cap, slab_bytes))) return rc;
  uint8_t* ph = static_cast<uint8_t*>(arena->plan_host);
  plan.fill(ph);
  std::memcpy(ph + off_copies, copies.data(), copies.size() * sizeof(CopyItem));
  HIP_TRY(hipMemcpyAsync(arena->plan_dev, arena->plan_host, plan.bytes, hipMemcpyHostToDevice, s));
  const uint8_t* pd = static_cast<const uint8_t*>(arena->plan_dev);
  plan.launch(pd, s);
  launch_copy_items(reinterpret_cast<const CopyItem*>(pd + off_copies), n, static_cast<uint8_t*>(arena->slabs_dev), s);
  HIP_TRY(hipMemcpyAsync(arena->slabs_host, arena->slabs_dev, slab_bytes, hipMemcpyDeviceToHost, s));
  for (uint32_t k = 0; k < n; ++k) jobs[which[k]]->readback = static_cast<uint8_t*>(arena->slabs_host) + copies[k].dst_offset;
  HIP_TRY(hipStreamSynchronize(s));
  return DMI_OK;
}

// Record prep of a whole batch (after every job's tables were normalised on the host, in plan mode): the coding tables of all
// jobs travel in one copy and are scattered to their buffers by one kernel; then one launch per prep kernel.
static int run_phase_b_batched(dmi_job** jobs, const std::vector<uint32_t>& which, const std::vector<std::vector<KernelStep>>& steps, BatchArena* arena, hipStream_t s) {
  if (which.empty()) return DMI_OK;
  BatchPlan plan;
  plan.add(steps, kPrepLevels);
  std::vector<CopyItem> items;
  size_t table_bytes = 0;
  for (uint32_t j : which)
    for (const auto& p : jobs[j]->run.pending) { items.push_back(CopyItem{p.dst, 0, (uint64_t)p.bytes}); table_bytes += (p.bytes + 255) & ~(size_t)255; }
  const size_t off_items = plan.reserve(items.size() * sizeof(CopyItem));
  const size_t off_tables = plan.reserve(table_bytes);
  int rc;
  if ((rc = BatchArena::reserve_pair(arena->plan_host, arena->plan_dev, arena->plan_cap, plan.bytes))) return rc;
  uint8_t* ph = static_cast<uint8_t*>(arena->plan_host);
  plan.fill(ph);
  {
    size_t at = off_tables, k = 0;
    for (uint32_t j : which)
      for (const auto& p : jobs[j]->run.pending) {
        std::memcpy(ph + at, p.src, p.bytes);   // (sources are padded to whole 16-byte words by their owners)
        items[k++].dst_offset = at;
        at += (p.bytes + 255) & ~(size_t)255;
      }
  }
  std::memcpy(ph + off_items, items.data(), items.size() * sizeof(CopyItem));
  HIP_TRY(hipMemcpyAsync(arena->plan_dev, arena->plan_host, plan.bytes, hipMemcpyHostToDevice, s));
  const uint8_t* pd = static_cast<const uint8_t*>(arena->plan_dev);
  launch_scatter_items(reinterpret_cast<const CopyItem*>(pd + off_items), (uint32_t)items.size(), pd, s);
  plan.launch(pd, s);
  return DMI_OK;
}

// Device form of a batch: the phases, the table stage and the record prep of ALL jobs are planned together (one upload, one
// launch per (level, kernel)), the chain descriptors are written by k_tables, and the chains follow on the same stream — the
// host waits for the first time when everything has been coded.  Read-back: one packed arena (coded bytes + serialised tables),
// one table of {offset, length, error}, 128 scratch bytes per attribute.
// A batch is begun (plan, upload, launches: returns without waiting) and finished (wait, read back, splice) separately, so that
// two halves of a large batch can be in flight on two streams: the data-parallel kernels of the second half run under the
// chain launch of the first, which is latency-bound by its longest stream and leaves the vector units idle.
static int parallel_items(uint32_t n, uint32_t n_threads, int device, const std::function<int(uint32_t)>& fn) {
  n_threads = std::max(1u, std::min(n, n_threads));
  std::vector<int> rcs(n_threads, DMI_OK);
  std::vector<std::string> errs(n_threads);
  auto work = [&](uint32_t t) {
    if (hipSetDevice(device) != hipSuccess) { rcs[t] = DMI_ERR_HIP; errs[t] = "hipSetDevice"; return; }
    const uint32_t lo = (uint32_t)((uint64_t)n * t / n_threads), hi = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
    for (uint32_t k = lo; k < hi; ++k) { const int rc = fn(k); if (rc) { rcs[t] = rc; errs[t] = g_last_error; return; } }
  };
  if (n_threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (uint32_t t = 0; t < n_threads; ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
  return DMI_OK;
}

struct DeviceBatch {
  std::vector<dmi_job*> jobs;
  std::vector<dmi_buffer*> outs;
  BatchArena* arena = nullptr;
  hipStream_t s = nullptr;
  uint32_t n_threads = 1;
  int device = 0;
  std::vector<uint32_t> first_desc, first_att;
  uint32_t n_streams = 0, n_atts = 0, n_descs = 0;
  size_t launches = 0;
  bool sparse_chains = false;
  double t_plan = 0, t_wait = 0, t_bytes = 0, t_splice = 0;
  ~DeviceBatch() {
    if (!arena) return;
    if (s) (void)hipStreamSynchronize(s);   // (also on error paths: nothing of this batch may still be writing into the arena when the next one takes it)
    release_batch_arena(arena);
  }

  int begin() {
    const uint32_t n = (uint32_t)jobs.size();
    int rc;
    const auto t0 = std::chrono::steady_clock::now();
    first_desc.assign(n + 1, 0);
    first_att.assign(n + 1, 0);
    for (uint32_t j = 0; j < n; ++j) { first_desc[j + 1] = first_desc[j] + count_streams(jobs[j]); first_att[j + 1] = first_att[j] + (uint32_t)jobs[j]->atts.size(); }
    n_streams = first_desc[n]; n_atts = first_att[n]; n_descs = n_streams + n_atts;
    const size_t order_at = (size_t)n_descs * sizeof(ChainDesc), counter_at = order_at + (((size_t)n_streams * 4 + 15) & ~(size_t)15);
    if ((rc = arena->reserve_descs(counter_at + 16))) return rc;
    ChainDesc* descs_dev = static_cast<ChainDesc*>(arena->descs_dev);
    // ---- plan (host threads; no HIP call) ----
    std::vector<std::vector<KernelStep>> steps(n);
    if ((rc = parallel_items(n, n_threads, device, [&](uint32_t j) {
          dmi_job* job = jobs[j];
          job->readback = nullptr;
          set_step_sink(&steps[j]);
          int r = encode_phase_a(job, true);
          const size_t n_a = steps[j].size();
          if (!r) r = encode_phase_b_dev(job, descs_dev + first_desc[j], descs_dev + n_streams + first_att[j]);
          set_step_sink(nullptr);
          for (size_t k = n_a; k < steps[j].size(); ++k) steps[j][k].level += kStepLevels;
          return r;
        }))) return rc;
    BatchPlan plan;
    plan.add(steps, kStepLevels + kPrepLevels);
    launches = plan.groups.size();
    // stream order for the chain kernel (longest first), scratch-word copies, capacities
    std::vector<uint64_t> length(n_streams);
    size_t cap_sum = 0;
    for (uint32_t j = 0; j < n; ++j)
      for (size_t k = 0; k < jobs[j]->run.descs.size(); ++k) { const ChainDesc& d = jobs[j]->run.descs[k]; length[first_desc[j] + k] = d.n; cap_sum += ((size_t)d.cap + 31) & ~(size_t)15; }
    for (uint32_t j = 0; j < n; ++j) for (auto& a : jobs[j]->atts) cap_sum += ((size_t)a.hdr_cap + 31) & ~(size_t)15;
    std::vector<uint32_t> by_length(n_streams);
    for (uint32_t k = 0; k < n_streams; ++k) by_length[k] = k;
    std::stable_sort(by_length.begin(), by_length.end(), [&](uint32_t x, uint32_t y) { return length[x] > length[y]; });
    std::vector<CopyItem> copies;
    copies.reserve(n_atts);
    for (uint32_t j = 0; j < n; ++j) for (auto& a : jobs[j]->atts) copies.push_back(CopyItem{a.small.p, (uint64_t)copies.size() * 128u, 128u});
    const size_t off_copies = plan.reserve(copies.size() * sizeof(CopyItem));
    const size_t off_order = plan.reserve((size_t)n_streams * 4);
    if ((rc = BatchArena::reserve_pair(arena->plan_host, arena->plan_dev, arena->plan_cap, plan.bytes))) return rc;
    if ((rc = BatchArena::reserve_pair(arena->slabs_host, arena->slabs_dev, arena->slabs_cap, (size_t)n_atts * 128))) return rc;
    if ((rc = arena->reserve(cap_sum, (size_t)(n_descs + 1) * sizeof(PackEntry)))) return rc;
    uint8_t* ph = static_cast<uint8_t*>(arena->plan_host);
    plan.fill(ph);
    std::memcpy(ph + off_copies, copies.data(), copies.size() * sizeof(CopyItem));
    std::memcpy(ph + off_order, by_length.data(), (size_t)n_streams * 4);
    // ---- the whole encode: one upload, then launches only ----
    HIP_TRY(hipMemcpyAsync(arena->plan_dev, arena->plan_host, plan.bytes, hipMemcpyHostToDevice, s));
    const uint8_t* pd = static_cast<const uint8_t*>(arena->plan_dev);
    plan.launch(pd, s);
    {
      uint64_t longest = 0, total = 0;
      for (uint64_t v : length) { longest = std::max(longest, v); total += v; }
      sparse_chains = chain_launch_sparse(longest, total, n_streams);
      launch_chains(descs_dev, reinterpret_cast<const uint32_t*>(pd + off_order), n_streams, reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(arena->descs_dev) + counter_at), sparse_chains, s);
    }
    launch_pack_streams(descs_dev, n_descs, static_cast<PackEntry*>(arena->table_dev), static_cast<uint8_t*>(arena->bytes_dev), s);
    launch_copy_items(reinterpret_cast<const CopyItem*>(pd + off_copies), n_atts, static_cast<uint8_t*>(arena->slabs_dev), s);
    HIP_TRY(hipMemcpyAsync(arena->table_host, arena->table_dev, (size_t)(n_descs + 1) * sizeof(PackEntry), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(arena->slabs_host, arena->slabs_dev, (size_t)n_atts * 128, hipMemcpyDeviceToHost, s));
    t_plan = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DMI_OK;
  }

  int finish() {
    const uint32_t n = (uint32_t)jobs.size();
    int rc;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t1 = now();
    HIP_TRY(hipStreamSynchronize(s));
    const auto t2 = now();
    const PackEntry* table = static_cast<const PackEntry*>(arena->table_host);
    const size_t total = (size_t)table[n_descs].offset;
    if ((rc = arena->reserve_host(total))) return rc;
    if (total) HIP_TRY(hipMemcpyAsync(arena->bytes_host, arena->bytes_dev, total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const auto t3 = now();
    const uint8_t* bytes_host = static_cast<const uint8_t*>(arena->bytes_host);
    if ((rc = parallel_items(n, n_threads, device, [&](uint32_t j) {
          dmi_job* job = jobs[j];
          const uint32_t na = (uint32_t)job->atts.size();
          job->readback = static_cast<uint8_t*>(arena->slabs_host) + (size_t)first_att[j] * 128;
          job->run.pin_off.assign(na, 0);
          job->run.hdr_ptr.assign(na, nullptr);
          job->run.hdr_len.assign(na, 0);
          for (uint32_t i = 0; i < na; ++i) {
            job->run.pin_off[i] = (size_t)i * 128;
            const uint32_t* small = reinterpret_cast<const uint32_t*>(job->readback + job->run.pin_off[i]);
            int frc = check_device_flags(small, i);
            if (!frc) frc = check_value_bounds(job->atts[i], small, i);
            if (frc) return frc;
            job->run.aux[i].zero_prob = (uint8_t)small[14];
            job->run.aux[i].count = small[15];
            const PackEntry& h = table[n_streams + first_att[j] + i];
            job->run.hdr_ptr[i] = bytes_host + h.offset;
            job->run.hdr_len[i] = h.len;
          }
          int r = encode_phase_c_packed(job, table, first_desc[j], bytes_host);
          if (!r) r = encode_phase_c3(job, outs[j]);
          return r;
        }))) return rc;
    t_wait = ms(t1, t2); t_bytes = ms(t2, t3); t_splice = ms(t3, now());
    return DMI_OK;
  }

  void trace(const char* name) const {   // per-stream chain clocks (100 MHz ticks written by the emitters) + host stages
    double sum_ms = 0, max_ms = 0, steps = 0, big_steps = 0, big_ms = 0;
    for (dmi_job* job : jobs)
      for (uint32_t i = 0; i < (uint32_t)job->atts.size(); ++i) {
        const uint32_t* small = reinterpret_cast<const uint32_t*>(job->readback + job->run.pin_off[i]);
        const double r = small[12] * 1e-5, x = job->run.aux[i].desc >= 0 ? small[13] * 1e-5 : 0.0;
        const double ns = (double)job->atts[i].n_sym, nx = job->run.aux[i].desc >= 0 ? (double)job->run.aux[i].count : 0.0;
        sum_ms += r + x; max_ms = std::max({max_ms, r, x}); steps += ns + nx;
        if (ns > 50000) { big_steps += ns; big_ms += r; }
      }
    std::fprintf(stderr, "[dmi] %s: %zu jobs, %u streams (%s chain launch), %zu launches; plan + issue %.2f ms, wait for the stream %.2f, byte read-back %.2f, splice %.2f; chains: %.0f steps, stream times sum %.1f ms "
                 "(/1024 walkers = %.2f), longest %.2f ms, %.1f ns/step (%.1f on rANS streams > 50k symbols)\n", name, jobs.size(), n_streams, sparse_chains ? "sparse" : "dense", launches, t_plan, t_wait, t_bytes, t_splice, steps, sum_ms,
                 sum_ms / 1024.0, max_ms, sum_ms * 1e6 / std::max(1.0, steps), big_ms * 1e6 / std::max(1.0, big_steps));
  }
};

// The two streams of a split batch (process lifetime, one pair per device; created back to back so that they land on different
// hardware queues — two streams that share a queue run their kernels strictly one after the other).
static bool batch_stream_pair(int device, hipStream_t& a, hipStream_t& b) {
  struct Pair { int device; hipStream_t a, b; };
  static std::mutex m;
  static std::vector<Pair> pairs;
  std::lock_guard<std::mutex> lock(m);
  for (auto& e : pairs) if (e.device == device) { a = e.a; b = e.b; return true; }
  Pair p{device, nullptr, nullptr};
  if (hipStreamCreateWithFlags(&p.a, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&p.b, hipStreamNonBlocking) != hipSuccess) return false;
  pairs.push_back(p);
  a = p.a; b = p.b;
  return true;
}

static int jobs_encode_device(dmi_job** jobs, uint32_t n, dmi_buffer* outs, uint32_t n_threads, bool trace) {
  const int device = jobs[0]->cfg.device;
  // Large batches run as two halves in flight: the jobs with the longest streams first (≈ 45 % of the symbols), the rest behind
  // them on a second stream — its data-parallel kernels (and the host's planning of it) run under the first half's chain launch.
  std::vector<uint32_t> order(n);
  for (uint32_t j = 0; j < n; ++j) order[j] = j;
  auto symbols = [&](uint32_t j) { uint64_t t = 0; for (auto& a : jobs[j]->atts) t += a.n_sym; return t; };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return symbols(x) > symbols(y); });
  uint64_t total = 0;
  for (uint32_t j = 0; j < n; ++j) total += symbols(j);
  uint32_t n_first = n;
  hipStream_t main_stream = jobs[0]->stream, side = nullptr;
  bool library_streams = true;   // a caller's stream (dmi_config.stream) is honoured: everything stays on it
  for (uint32_t j = 0; j < n; ++j) if (jobs[j]->cfg.stream) library_streams = false;
  if (n >= 32 && library_streams && std::getenv("DMI_SPLIT") && batch_stream_pair(device, main_stream, side)) {
    uint64_t acc = 0;
    n_first = 0;
    while (n_first < n && acc * 100 < total * 45) acc += symbols(order[n_first++]);
    if (n_first < 8 || n - n_first < 8) n_first = n;
  }
  DeviceBatch first, second;
  auto fill = [&](DeviceBatch& b, uint32_t lo, uint32_t hi, hipStream_t s) {
    for (uint32_t k = lo; k < hi; ++k) { b.jobs.push_back(jobs[order[k]]); b.outs.push_back(&outs[order[k]]); }
    b.arena = acquire_batch_arena(device);
    b.s = s; b.n_threads = n_threads; b.device = device;
  };
  fill(first, 0, n_first, main_stream);
  int rc;
  if ((rc = first.begin())) return rc;
  if (n_first < n) {
    fill(second, n_first, n, side);
    if ((rc = second.begin())) { (void)hipStreamSynchronize(first.s); (void)hipStreamSynchronize(second.s); return rc; }
  }
  rc = first.finish();
  if (n_first < n) {
    if (rc) (void)hipStreamSynchronize(second.s);   // a failed first half: let the second drain, report the first error
    else rc = second.finish();
  }
  if (rc) return rc;
  if (trace) { first.trace(n_first < n ? "batch, device form, first half" : "batch, device form"); if (n_first < n) second.trace("batch, device form, second half"); }
  return DMI_OK;
}

static int jobs_encode_impl(dmi_job** jobs, uint32_t n, dmi_buffer* outs);
int dmi_jobs_encode(dmi_job** jobs, uint32_t n, dmi_buffer* outs) {
  if (!jobs || !outs || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  for (uint32_t j = 0; j < n; ++j) outs[j] = dmi_buffer{};
  const int rc = jobs_encode_impl(jobs, n, outs);
  if (rc) {   // all or nothing: no output of a failed batch is left allocated (the error text survives the frees)
    const std::string why = g_last_error;
    dmi_free_many(outs, n);
    g_last_error = why;
  }
  return rc;
}
// One process, several GPUs: the jobs are grouped by the device they live on and every group is coded by its own dmi_jobs_encode
// on its own host thread — the devices run concurrently, the call returns when all have finished.  All or nothing.
int dmi_jobs_encode_devices(dmi_job** jobs, uint32_t n, dmi_buffer* outs) {
  if (!jobs || !outs || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  for (uint32_t j = 0; j < n; ++j) { outs[j] = dmi_buffer{}; if (!jobs[j]) return fail(DMI_ERR_INVALID_ARGUMENT, "null job"); }
  std::vector<int> devices;
  for (uint32_t j = 0; j < n; ++j) if (std::find(devices.begin(), devices.end(), jobs[j]->cfg.device) == devices.end()) devices.push_back(jobs[j]->cfg.device);
  if (devices.size() == 1) return dmi_jobs_encode(jobs, n, outs);
  std::vector<int> rcs(devices.size(), DMI_OK);
  std::vector<std::string> errs(devices.size());
  auto work = [&](size_t g) {
    std::vector<dmi_job*> mine;
    std::vector<uint32_t> at;
    for (uint32_t j = 0; j < n; ++j) if (jobs[j]->cfg.device == devices[g]) { mine.push_back(jobs[j]); at.push_back(j); }
    std::vector<dmi_buffer> got(mine.size());
    rcs[g] = dmi_jobs_encode(mine.data(), (uint32_t)mine.size(), got.data());
    if (rcs[g]) { errs[g] = g_last_error; return; }
    for (size_t k = 0; k < at.size(); ++k) outs[at[k]] = got[k];
  };
  std::vector<std::thread> th;
  for (size_t g = 1; g < devices.size(); ++g) th.emplace_back(work, g);
  work(0);
  for (auto& x : th) x.join();
  for (size_t g = 0; g < devices.size(); ++g)
    if (rcs[g]) {
      dmi_free_many(outs, n);
      return fail(rcs[g], "device " + std::to_string(devices[g]) + ": " + errs[g]);
    }
  return DMI_OK;
}
static int jobs_encode_impl(dmi_job** jobs, uint32_t n, dmi_buffer* outs) {
  for (uint32_t j = 0; j < n; ++j) if (!jobs[j] || jobs[j]->cfg.device != jobs[0]->cfg.device) return fail(DMI_ERR_INVALID_ARGUMENT, "batched jobs must live on one device");
  const int device = jobs[0]->cfg.device;
  // Small meshes are launch-bound, so whatever stays per job (table normalisation; the phases of jobs that cannot be planned ahead) runs on several host
  // threads, each walking a contiguous slice of the jobs (jobs that own their stream then also overlap on the GPU).
  static const uint32_t thread_cap = std::getenv("DMI_BATCH_THREADS") ? (uint32_t)std::atoi(std::getenv("DMI_BATCH_THREADS")) : 16u;
  const uint32_t n_threads = std::max(1u, std::min({n, (uint32_t)host_threads(), std::max(1u, thread_cap)}));
  auto parallel = [&](auto&& fn, bool sync_after = true) -> int {
    std::vector<int> rcs(n_threads, DMI_OK);
    std::vector<std::string> errs(n_threads);
    auto work = [&](uint32_t t) {
      if (hipSetDevice(device) != hipSuccess) { rcs[t] = DMI_ERR_HIP; errs[t] = "hipSetDevice"; return; }
      const uint32_t lo = (uint32_t)((uint64_t)n * t / n_threads), hi = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
      for (uint32_t j = lo; j < hi; ++j) { const int rc = fn(j); if (rc) { rcs[t] = rc; errs[t] = g_last_error; return; } }
      if (sync_after) {   // each worker waits for its own slice's streams
        hipStream_t last = nullptr;
        for (uint32_t j = lo; j < hi; ++j) {
          if (j > lo && jobs[j]->stream == last) continue;
          last = jobs[j]->stream;
          if (hipStreamSynchronize(last) != hipSuccess) { rcs[t] = DMI_ERR_HIP; errs[t] = "hipStreamSynchronize"; return; }
        }
      }
    };
    if (n_threads == 1) work(0);
    else {
      std::vector<std::thread> th;
      for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(work, t);
      for (auto& x : th) x.join();
    }
    for (uint32_t t = 0; t < n_threads; ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
    return DMI_OK;
  };
  hipStream_t s = jobs[0]->stream;
  int rc;
  const bool trace = std::getenv("DMI_TRACE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  // Pipeline per worker: phase A of every job of its share is queued first; then, job by job, the worker waits for
  // that job's histograms, normalises its tables and queues the record prep — host work of early jobs overlaps the
  // data-parallel kernels of later ones.  The chains of ALL jobs then run in one launch: a long-running kernel per job
  // would pin one of the few hardware queues each and serialise the batch (measured: 114 ms instead of 8).
  BatchArena* arena = acquire_batch_arena(device);
  struct Release { BatchArena* a; ~Release() { if (a) release_batch_arena(a); } } release{arena};
  // jobs whose phase A can be planned ahead (no mid-phase host wait, no per-job event timing) share one launch per kernel
  std::vector<uint32_t> batched;
  std::vector<uint8_t> is_batched(n, 0);
  for (uint32_t j = 0; j < n; ++j) {
    jobs[j]->readback = nullptr;
    bool ok = !jobs[j]->have_events && !std::getenv("DMI_NO_BATCHED_PHASES");
    for (auto& a : jobs[j]->atts) if (a.port == kToBits) ok = false;
    if (ok) { batched.push_back(j); is_batched[j] = 1; }
  }
  HIP_TRY(hipSetDevice(device));
  {
    bool all_device = batched.size() == n;
    for (uint32_t j = 0; j < n && all_device; ++j) all_device = jobs[j]->dev_tables;
    if (all_device) { release.a = nullptr; release_batch_arena(arena); return jobs_encode_device(jobs, n, outs, n_threads, trace); }
  }
  if ((rc = run_phase_a_batched(jobs, batched, arena, s))) return rc;
  const auto t1 = now();
  std::vector<std::vector<KernelStep>> b_steps(n);   // record-prep steps of the batched jobs (filled by the workers)
  std::vector<uint32_t> order(n);
  for (uint32_t j = 0; j < n; ++j) order[j] = j;
  auto job_size = [&](uint32_t j) { uint64_t t = 0; for (auto& a : jobs[j]->atts) t += a.n_sym; return t; };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return job_size(x) > job_size(y); });
  {
    std::vector<int> rcs(n_threads, DMI_OK);
    std::vector<std::string> errs(n_threads);
    auto work = [&](uint32_t t) {
      auto bail = [&](int rc_, const std::string& e) { rcs[t] = rc_; errs[t] = e; };
      if (hipSetDevice(device) != hipSuccess) return bail(DMI_ERR_HIP, "hipSetDevice");
      // jobs dealt round-robin in descending size: every worker gets a mix, its first job is one of the largest
      std::vector<uint32_t> mine;
      for (uint32_t k = t; k < n; k += n_threads) mine.push_back(order[k]);
      const auto w0 = std::chrono::steady_clock::now();
      for (uint32_t j : mine) { if (is_batched[j]) continue; const int r = run_phase_a(jobs[j]); if (r) return bail(r, g_last_error); }
      const auto w1 = std::chrono::steady_clock::now();
      double wait_ms = 0, b_ms = 0;
      for (uint32_t j : mine) {
        dmi_job* job = jobs[j];
        const auto x0 = std::chrono::steady_clock::now();
        if (!is_batched[j] && hipStreamSynchronize(job->stream) != hipSuccess) return bail(DMI_ERR_HIP, "hipStreamSynchronize");
        const auto x1 = std::chrono::steady_clock::now();
        int r;
        if (is_batched[j]) { set_step_sink(&b_steps[j]); r = encode_phase_b(job, true); set_step_sink(nullptr); }
        else r = encode_phase_b(job);
        if (r) return bail(r, g_last_error);
        const auto x2 = std::chrono::steady_clock::now();
        wait_ms += std::chrono::duration<double, std::milli>(x1 - x0).count();
        b_ms += std::chrono::duration<double, std::milli>(x2 - x1).count();
      }
      const auto w2 = std::chrono::steady_clock::now();
      for (uint32_t j : mine) if (hipStreamSynchronize(jobs[j]->stream) != hipSuccess) return bail(DMI_ERR_HIP, "hipStreamSynchronize");
      if (trace && t == 0) std::fprintf(stderr, "[dmi] worker 0: %zu jobs, phase A issue %.2f ms, waits for histograms %.2f, phase B host+issue %.2f, final wait %.2f\n", mine.size(),
                                        std::chrono::duration<double, std::milli>(w1 - w0).count(), wait_ms, b_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w2).count());
    };
    if (n_threads == 1) work(0);
    else {
      std::vector<std::thread> th;
      for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(work, t);
      for (auto& x : th) x.join();
    }
    for (uint32_t t = 0; t < n_threads; ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
  }
  const auto t3 = now();
  {
    std::vector<std::vector<KernelStep>> only;
    only.reserve(batched.size());
    for (uint32_t j : batched) only.push_back(std::move(b_steps[j]));
    if ((rc = run_phase_b_batched(jobs, batched, only, arena, s))) return rc;
  }
  const auto t4 = now();
  std::vector<ChainDesc> all;
  for (uint32_t j = 0; j < n; ++j) all.insert(all.end(), jobs[j]->run.descs.begin(), jobs[j]->run.descs.end());
  // the chain kernel serves the streams longest first (its pairs pull work; see k_chains)
  std::vector<uint32_t> by_length(all.size());
  for (uint32_t k = 0; k < (uint32_t)all.size(); ++k) by_length[k] = k;
  std::stable_sort(by_length.begin(), by_length.end(), [&](uint32_t x, uint32_t y) { return all[x].n > all[y].n; });
  DevMem descs_dev;   // descriptors | order | pull counter
  const size_t order_at = all.size() * sizeof(ChainDesc), counter_at = order_at + ((all.size() * sizeof(uint32_t) + 15) & ~(size_t)15);
  if ((rc = descs_dev.alloc(counter_at + 16))) return rc;
  HIP_TRY(hipMemcpyAsync(descs_dev.p, all.data(), order_at, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(descs_dev.p) + order_at, by_length.data(), all.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  {
    uint64_t longest = 0, total = 0;
    for (const ChainDesc& cd : all) { longest = std::max<uint64_t>(longest, cd.n); total += cd.n; }
    launch_chains(descs_dev.as<ChainDesc>(), reinterpret_cast<const uint32_t*>(static_cast<uint8_t*>(descs_dev.p) + order_at), (uint32_t)all.size(),
                  reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(descs_dev.p) + counter_at), chain_launch_sparse(longest, total, (uint32_t)all.size()), s);
  }
  HIP_TRY(hipStreamSynchronize(s));
  const auto t5 = now();
  // read-back: every stream of every job packed into one arena on the device → one table copy + one byte copy
  const uint32_t n_streams = (uint32_t)all.size();
  std::vector<uint32_t> first_desc(n, 0);
  size_t cap_sum = 0;
  {
    uint32_t at = 0;
    for (uint32_t j = 0; j < n; ++j) { first_desc[j] = at; at += (uint32_t)jobs[j]->run.descs.size(); }
    for (const ChainDesc& d : all) cap_sum += ((size_t)d.cap + 31) & ~(size_t)15;
  }
  if ((rc = arena->reserve(cap_sum, (size_t)(n_streams + 1) * sizeof(PackEntry)))) return rc;
  launch_pack_streams(descs_dev.as<ChainDesc>(), n_streams, static_cast<PackEntry*>(arena->table_dev), static_cast<uint8_t*>(arena->bytes_dev), s);
  HIP_TRY(hipMemcpyAsync(arena->table_host, arena->table_dev, (size_t)(n_streams + 1) * sizeof(PackEntry), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const PackEntry* table = static_cast<const PackEntry*>(arena->table_host);
  const size_t total = (size_t)table[n_streams].offset;
  if ((rc = arena->reserve_host(total))) return rc;
  if (total) HIP_TRY(hipMemcpyAsync(arena->bytes_host, arena->bytes_dev, total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (uint32_t j = 0; j < n; ++j) if ((rc = encode_phase_c_packed(jobs[j], table, first_desc[j], static_cast<const uint8_t*>(arena->bytes_host)))) return rc;
  const auto t6 = now();
  if ((rc = parallel([&](uint32_t j) { return encode_phase_c3(jobs[j], &outs[j]); }, false))) return rc;
  if (trace) std::fprintf(stderr, "[dmi] batch of %u (%zu with batched phases) on %u host threads: data-parallel phases %.2f ms, tables (host threads) %.2f + record prep plan/upload/launch %.2f, chains (%zu streams, one launch) %.2f, packed read-back %.2f, splice %.2f\n", n, batched.size(), n_threads, ms(t0, t1), ms(t1, t3), ms(t3, t4), all.size(), ms(t4, t5), ms(t5, t6), ms(t6, now()));
  return DMI_OK;
}

int dmi_encode_attributes(const dmi_attribute* atts, const dmi_corner_table* tables, uint32_t n_atts, const uint32_t* seeds, uint32_t n_seeds,
                          const dmi_config* cfg, dmi_buffer* out) {
  dmi_job* job = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  int rc = dmi_job_create(atts, tables, n_atts, seeds, n_seeds, cfg, &job);
  if (rc) return rc;
  const float t_create = ms();
  rc = dmi_job_encode(job, out);
  g_last_call = job->last;
  g_last_call.job_create_ms = t_create;
  dmi_job_destroy(job);
  g_last_call.call_ms = ms();
  return rc;
}

int dmi_encode_attributes_batch(const dmi_batch_item* items, uint32_t n, const dmi_config* cfg_in, dmi_buffer* outs) {
  if (!items || !outs || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  dmi_config cfg{};
  if (cfg_in) cfg = *cfg_in;
  HIP_TRY(hipSetDevice(cfg.device));
  hipStream_t own = nullptr;
  if (!cfg.stream) { HIP_TRY(hipStreamCreate(&own)); cfg.stream = own; }
  std::vector<dmi_job*> jobs(n, nullptr);
  int rc = DMI_OK;
  for (uint32_t j = 0; j < n && !rc; ++j) rc = dmi_job_create(items[j].atts, items[j].tables, items[j].n_atts, items[j].seeds, items[j].n_seeds, &cfg, &jobs[j]);
  if (!rc) rc = dmi_jobs_encode(jobs.data(), n, outs);
  for (auto* j : jobs) dmi_job_destroy(j);
  if (own) (void)hipStreamDestroy(own);
  return rc;
}

}  // extern "C"
