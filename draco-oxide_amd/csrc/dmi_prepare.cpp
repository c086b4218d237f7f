// dmi_prepare.cpp — whole-mesh entry points of the C ABI: the connectivity stage (host walks + device tables), dmi_mesh_prepare /
// dmi_meshes_prepare / dmi_encode_mesh, and the host-core stream coders on their own.
#include "dmi_job.hpp"

using namespace dmi;

#include <sched.h>

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

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
    static const bool off = std::getenv("DMI_NO_NUMA_PIN") != nullptr;
    if (off) return;
    const int node = device_numa_node(device);
    if (node < 0) return;
    cpu_set_t want;
    CPU_ZERO(&want);
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    std::FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return;
    char buf[1024] = {0};
    const bool got = std::fgets(buf, sizeof buf, f) != nullptr;
    std::fclose(f);
    if (!got) return;
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
std::shared_ptr<StreamHolder> thread_stream(int device) { return thread_streams().get(0, device); }

// The universal corner table of a mesh as the device connectivity stage built it, read back for the host's serial walks (pinned staging).
struct PrebuiltTable {
  const uint32_t *c2v = nullptr, *opp = nullptr, *lmc = nullptr;
  const uint8_t* on_boundary = nullptr;
  uint32_t V = 0;
  bool no_boundary = false;
  // attribute corner tables the device built (k_att_*), by index among the mesh's non-position attributes; ready = false: the host builds it
  struct Att { bool ready = false, interior = false; uint32_t nv = 0; const uint8_t* seam = nullptr; const uint32_t *c2v = nullptr, *opp = nullptr, *lmc = nullptr;
               const uint32_t *d_c2v = nullptr, *d_opp = nullptr; };
  std::vector<Att> att;
};

constexpr uint32_t kDeviceTablesMinFaces = 1u << 16;   // a single mesh from this size up gets its universal corner table from the device (dmi_conn.hip)

// The universal corner table of ONE mesh built on the device (dmi_conn.hip) and read back for the host's serial walks: faces up,
// opposite corners + per-vertex boundary flags down (pinned staging), the device copies kept for job creation (coding-order relabelling
// reads them where they are).  Meshes the order-free construction does not cover (flags) take the host builder instead.
struct DeviceTables {
  int device = 0;
  hipStream_t stream = nullptr;
  TempDev mem;
  HostStage* host = nullptr;
  uint32_t *d_faces = nullptr, *d_c2v = nullptr, *d_opp = nullptr, *d_lmc = nullptr;
  uint8_t* d_onb = nullptr;
  PrebuiltTable pre;
  uint32_t V = 0, Vcap = 0, flags = 0;
  bool valid = false;
  double t_up = 0, t_kernels = 0, t_down = 0;
  ~DeviceTables() { if (stream && valid) (void)hipStreamSynchronize(stream); release_stage(host); }

  // DMI_OK with valid = true: ct views the tables; DMI_OK with valid = false: not covered (the caller runs the host builder); else an error
  // src_faces / src_pos_map (nullable): the mesh's faces / position map already in device memory (dmi_encode_mesh_device) — used where they
  // are; `mesh` then holds their host copies on huge pages (the staging they were read back into)
  int build(const dmi_mesh* mesh, std::vector<uint32_t>& c2v_store, const uint32_t* src_faces = nullptr, const uint32_t* src_pos_map = nullptr) {
    const uint32_t F = mesh->num_faces;
    const size_t C = (size_t)F * 3;
    const dmi_attribute& pos = mesh->atts[0];
    Vcap = pos.num_unique;
    const uint32_t P = pos.num_points;
    if (!F || !Vcap || Vcap >= (1u << 31) || C >= (1ull << 32)) return DMI_OK;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    HIP_TRY(hipSetDevice(device));
    const bool mapped = pos.point_to_value != nullptr;
    const size_t nv = (size_t)Vcap + 1, parts = scan_partials_words((uint32_t)nv);
    mem.init(device, stream, C * 4 * (mapped ? 5 : 4) + (mapped ? (size_t)P * 4 : 0) + nv * 4 * 4 + nv + C + parts * 4 + ((size_t)1 << 16));
    d_faces = src_faces ? const_cast<uint32_t*>(src_faces) : mem.take<uint32_t>(C);
    uint32_t* d_p2v = mapped ? (src_pos_map ? const_cast<uint32_t*>(src_pos_map) : mem.take<uint32_t>(P)) : nullptr;
    d_c2v = mapped ? mem.take<uint32_t>(C) : d_faces;
    d_opp = mem.take<uint32_t>(C);
    d_lmc = mem.take<uint32_t>(nv);
    d_onb = mem.take<uint8_t>(nv);
    ConnArgs a{};
    a.ecount = mem.take<uint32_t>(nv); a.efill = mem.take<uint32_t>(nv); a.first = mem.take<uint32_t>(nv);
    a.he_key = mem.take<uint32_t>(C); a.he_corner = mem.take<uint32_t>(C);
    a.cdone = mem.take<uint8_t>(C);
    a.scan_partials = mem.take<uint32_t>(parts);
    ConnMeshDesc* d_desc = mem.take<ConnMeshDesc>(1);
    uint32_t* d_words = mem.take<uint32_t>(4);
    if (!d_faces || (mapped && (!d_p2v || !d_c2v)) || !d_opp || !d_lmc || !d_onb || !a.ecount || !a.efill || !a.first || !a.he_key || !a.he_corner || !a.cdone || !a.scan_partials || !d_desc || !d_words)
      return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (device connectivity tables)");
    const size_t host_need = C * 4 * (mapped ? 2 : 1) + nv * 5 + 1024;
    host = acquire_stage(device, host_need);
    if (!host) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (connectivity read-back)");
    uint8_t* hp = host->p;
    uint32_t* hp_opp = reinterpret_cast<uint32_t*>(hp);
    uint32_t* hp_c2v = mapped ? hp_opp + C : nullptr;
    uint32_t* hp_lmc = reinterpret_cast<uint32_t*>(hp + C * 4 * (mapped ? 2 : 1));
    uint8_t* hp_onb = reinterpret_cast<uint8_t*>(hp_lmc + nv);
    uint32_t* hp_words = reinterpret_cast<uint32_t*>(hp + ((C * 4 * (mapped ? 2 : 1) + nv * 5 + 255) & ~(size_t)255));
    const ConnMeshDesc desc{0u, 0u, F, Vcap, mapped ? 0u : kNone, P, 0u, 0u};
    HIP_TRY(hipMemcpyAsync(d_desc, &desc, sizeof desc, hipMemcpyHostToDevice, stream));
    if (!src_faces) HIP_TRY(hipMemcpyAsync(d_faces, mesh->faces, C * 4, hipMemcpyHostToDevice, stream));
    if (mapped && !src_pos_map) HIP_TRY(hipMemcpyAsync(d_p2v, pos.point_to_value, (size_t)P * 4, hipMemcpyHostToDevice, stream));
    t_up = ms();
    a.meshes = d_desc; a.M = 1; a.total_faces = F; a.total_verts = Vcap;
    a.faces = d_faces; a.p2v = d_p2v; a.c2v = d_c2v; a.opp = d_opp; a.lmc = d_lmc; a.on_boundary = d_onb; a.flags = d_words; a.vmax = d_words + 1;
    HIP_TRY(conn_tables_clear(a, stream));
    launch_conn_tables(a, stream);
    HIP_TRY(hipMemcpyAsync(hp_words, d_words, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(hp_opp, d_opp, C * 4, hipMemcpyDeviceToHost, stream));
    if (mapped) HIP_TRY(hipMemcpyAsync(hp_c2v, d_c2v, C * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(hp_onb, d_onb, Vcap, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(hp_lmc, d_lmc, (size_t)Vcap * 4, hipMemcpyDeviceToHost, stream));
    t_kernels = ms();
    // while the device works: the vertex ids of a mesh without a position map are its faces — the walks read them at random, so they get a
    // copy on huge pages (the caller's array is on whatever pages its allocator chose); storage from the host pool, written in parallel slices
    const uint32_t* c2v_host = nullptr;
    if (!mapped && src_faces) c2v_host = mesh->faces;
    else if (!mapped) {
      pool_fit(c2v_store, C);
      if (c2v_store.capacity() < C) c2v_store.reserve(C);   // (below the pool's size threshold pool_fit hands out an empty vector)
      uint32_t* dst = c2v_store.data();   // (capacity ≥ C; the vector's size stays 0: it only carries the storage back to the pool)
      const uint32_t* src = mesh->faces;
      parallel_for(C, [&](size_t lo, size_t hi) { std::memcpy(dst + lo, src + lo, (hi - lo) * 4); });
      c2v_host = dst;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    t_down = ms();
    flags = hp_words[0];
    if (flags & CONN_BAD_INDEX) return fail(DMI_ERR_INVALID_ARGUMENT, "face index ≥ number of points, or a position value index out of range");
    if (flags & (CONN_DEGENERATE | CONN_NONMANIFOLD_EDGE | CONN_MULTI_FAN)) return DMI_OK;   // the reference's serial walks decide (host_conn.cpp)
    if (flags & CONN_UNUSED_VERTEX) return fail(DMI_ERR_UNUSED_VERTICES, "mesh contains unused vertices");
    V = hp_words[1] + 1;
    pre.c2v = mapped ? hp_c2v : c2v_host; pre.opp = hp_opp; pre.lmc = hp_lmc; pre.on_boundary = hp_onb; pre.V = V;
    pre.no_boundary = !(flags & CONN_HAS_BOUNDARY);
    valid = true;
    return DMI_OK;
  }
};
void view_prebuilt(CornerTables& ct, const dmi_mesh* mesh, const PrebuiltTable& pre) {
  ct.F = mesh->num_faces; ct.V = pre.V;
  ct.c2p = mesh->faces; ct.c2v = pre.c2v; ct.opp = pre.opp; ct.lmc = pre.lmc;
  ct.no_boundary = pre.no_boundary;
  ct.att.clear();
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------
// Whole-mesh entry points: connectivity stage (host walks, device tables) + device attributes.
// ------------------------------------------------------------------------------------------------
struct ConnOwner {
  CornerTables ct;
  EdgebreakerResult eb;
  std::vector<std::vector<uint32_t>> seqs;
  ConnOwner() = default;
  ConnOwner(const ConnOwner&) = delete;
  ConnOwner& operator=(const ConnOwner&) = delete;
  ~ConnOwner() {   // the large arrays go back to the host pool (dmi_host.hpp)
    pool_give(ct.c2p_own); pool_give(ct.c2v_own); pool_give(ct.opp_own); pool_give(ct.lmc_own);
    for (auto& a : ct.att) { pool_give(a.c2v); pool_give(a.opp); pool_give(a.lmc); pool_give(a.seam_edge); }
    pool_give(eb.seeds); pool_give(eb.processed);
    for (auto& q : seqs) pool_give(q);
  }
  std::vector<dmi_corner_table> views;
};

// thread time of the connectivity stage by step, summed over the meshes of a batch (trace): attribute tables, Edgebreaker (traversal +
// connectivity bytes incl. the seam streams), universal sequencer, seam-table sequencers
static std::atomic<uint64_t> g_conn_us[4];
// set by a batch worker whose batch keeps every host thread busy with a mesh of its own: a large mesh then walks its steps one after the other on
// its worker (the overlapped form starts three more threads per mesh — with 16 workers on 16 CPUs they only wait for each other)
static thread_local bool g_batch_worker_busy = false;
// dmi_encode_mesh_device's read-back of the faces / maps (on a stream of its own, beside the table kernels): waited for right before the host reads them
static thread_local hipEvent_t g_faces_event = nullptr;
// pre (nullable): the universal table already built by the device stage; view_faces: c2p may view the caller's face array (it outlives `o`)
static int build_connectivity(const dmi_mesh* mesh, ConnOwner& o, std::vector<uint8_t>& bytes, const PrebuiltTable* pre = nullptr, bool view_faces = false) {
  if (!mesh || !mesh->atts || mesh->num_atts == 0 || (!mesh->faces && mesh->num_faces)) return fail(DMI_ERR_INVALID_ARGUMENT, "bad mesh");
  if (mesh->atts[0].att_type != DMI_ATT_POSITION) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute 0 must be the Position attribute (core/mesh/builder.rs:115-125)");
  for (uint32_t i = 0; i < mesh->num_atts; ++i)
    if (mesh->atts[i].point_to_value == nullptr && mesh->atts[i].num_unique < mesh->atts[i].num_points) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + ": fewer values than points and no point_to_value map");
  std::string err;
  const bool trace = std::getenv("DMI_TRACE") != nullptr;
  auto tick = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  auto c0 = tick();
  int rc = DMI_OK;
  const bool on_device = pre != nullptr;   // (the device pass range-checked the faces and the position map itself)
  if (pre) view_prebuilt(o.ct, mesh, *pre);
  {   // faces index the attributes' points (and, through point_to_value, their values) from here on
    const size_t C = (size_t)mesh->num_faces * 3;
    const uint32_t P = mesh->atts[0].num_points;
    std::atomic<int> bad{0};
    if (!on_device) parallel_for(C, [&](size_t lo, size_t hi) { for (size_t c = lo; c < hi; ++c) if (mesh->faces[c] >= P) { bad.store(1); break; } });
    if (bad) return fail(DMI_ERR_INVALID_ARGUMENT, "face index ≥ number of points");
    for (uint32_t i = 0; i < mesh->num_atts; ++i) {
      const dmi_attribute& a = mesh->atts[i];
      if (a.num_points < P) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + " has fewer points than the Position attribute");
      if (a.point_to_value) parallel_for(a.num_points, [&](size_t lo, size_t hi) { for (size_t p = lo; p < hi; ++p) if (a.point_to_value[p] >= a.num_unique) { bad.store(1); break; } });
      if (bad) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + ": point_to_value entry out of range");
    }
  }
  if (!on_device) {
    rc = o.ct.build_universal(mesh->faces, mesh->num_faces, mesh->atts[0].point_to_value, err, /*copy_faces=*/!view_faces);
    if (rc) return fail(rc, err);
  }
  const double t_univ = since(c0);
  // The serial graph walks of one large mesh overlap on a few host threads: the attribute corner tables (their loops are parallel
  // themselves) are built while the Edgebreaker traversal runs (it reads the universal table only; the seam flags are needed at its
  // end), and the attribute sequencers start as soon as the traversal has produced its seeds, beside the assembly of the
  // connectivity bytes.  Small meshes (a batch already runs one mesh per host thread) do the same steps one after the other.
  std::vector<const uint32_t*> maps;
  for (uint32_t i = 0; i < mesh->num_atts; ++i) if (mesh->atts[i].att_type != DMI_ATT_POSITION) maps.push_back(mesh->atts[i].point_to_value);
  o.ct.att.resize(maps.size());
  const bool overlap = mesh->num_faces > 100000 && !g_batch_worker_busy;
  double t_att = 0, t_eb = 0, t_seq = 0;
  o.views.resize(mesh->num_atts);
  o.seqs.resize(mesh->num_atts);
  auto universal_view = [&](dmi_corner_table& v) { v.num_vertices = o.ct.V; v.corner_to_vertex = o.ct.c2v; v.opposite = o.ct.opp; v.left_most_corner = o.ct.lmc; };
  Pooled<uint8_t> on_boundary_p;
  std::vector<uint8_t>& on_boundary = on_boundary_p.v;
  const uint8_t* boundary_flags = on_device ? pre->on_boundary : nullptr;   // per vertex: on a boundary of the universal table (the device pass computes them with the left-most corners)
  auto sequence_universal = [&] {
    TableRef tr{o.ct.F, o.ct.V, o.ct.c2v, o.ct.opp, o.ct.lmc};
    attribute_sequence(tr, o.eb, o.seqs[0], boundary_flags ? boundary_flags : (on_boundary.empty() ? nullptr : on_boundary.data()));
  };

  auto build_att_tables = [&] {
    const auto a0 = tick();
    // an attribute indexed like the Position attribute has no seams but the boundary; one indexed like an earlier attribute has that one's table
    // an attribute indexed like the Position attribute — the same map array, or a map with the same entries (a builder gives every attribute its
    // own array: normals that repeat exactly where positions repeat) — has no seam but the boundary: one memcmp instead of a test per edge
    const uint32_t* pos_map = mesh->atts[0].point_to_value;
    const uint32_t P0 = mesh->atts[0].num_points;
    auto like_position = [&](const uint32_t* m, uint32_t n_points) {
      if (m == pos_map) return true;
      return m && pos_map && n_points == P0 && std::memcmp(m, pos_map, (size_t)P0 * 4) == 0;
    };
    std::vector<uint32_t> map_points;
    for (uint32_t i = 0; i < mesh->num_atts; ++i) if (mesh->atts[i].att_type != DMI_ATT_POSITION) map_points.push_back(mesh->atts[i].num_points);
    auto build_one = [&](size_t k) {
      for (size_t j = 0; j < k; ++j) if (maps[j] == maps[k]) return;   // (copied below, once its original is complete)
      if (pre && k < pre->att.size() && pre->att[k].ready) {   // built by the device (k_att_*) and read back: copied into the walks' arrays
        const PrebuiltTable::Att& pa = pre->att[k];
        AttTable& a = o.ct.att[k];
        if (!pa.interior) { o.ct.build_attribute_into(a, nullptr, /*same_as_position=*/true); return; }   // (no seam but the boundary: flags from the universal table)
        const size_t C = (size_t)o.ct.F * 3;
        a.alias_of = -1; a.interior_seams = true; a.num_vertices = pa.nv;
        pool_fit(a.seam_edge, C); a.seam_edge.assign(pa.seam, pa.seam + C);
        pool_fit(a.c2v, C); a.c2v.assign(pa.c2v, pa.c2v + C);
        pool_fit(a.opp, C); a.opp.assign(pa.opp, pa.opp + C);
        pool_fit(a.lmc, pa.nv); a.lmc.assign(pa.lmc, pa.lmc + pa.nv);
        return;
      }
      o.ct.build_attribute_into(o.ct.att[k], maps[k], like_position(maps[k], map_points[k]));
    };
    if (maps.size() > 1 && overlap) {
      std::vector<std::thread> th;
      for (size_t k = 0; k < maps.size(); ++k) th.emplace_back(build_one, k);
      for (auto& x : th) x.join();
    } else {
      for (size_t k = 0; k < maps.size(); ++k) build_one(k);
    }
    for (size_t k = 0; k < maps.size(); ++k)
      for (size_t j = 0; j < k; ++j) if (maps[j] == maps[k]) { o.ct.copy_attribute_into(o.ct.att[k], o.ct.att[j]); o.ct.att[k].alias_of = (int)j; break; }
    t_att = since(a0);
  };
  std::thread att_thread, seq_thread, flag_thread;
  EdgebreakerHooks hooks;
  if (overlap) {
    // the sequencer's per-vertex boundary test, ahead of time (beside the start of the traversal)
    if (!boundary_flags) flag_thread = std::thread([&] { TableRef tr{o.ct.F, o.ct.V, o.ct.c2v, o.ct.opp, o.ct.lmc}; vertex_boundary_flags(tr, on_boundary); });
    att_thread = std::thread(build_att_tables);
    hooks.seeds_ready = [&] { if (flag_thread.joinable()) flag_thread.join(); seq_thread = std::thread([&] { const auto q0 = tick(); sequence_universal(); t_seq = since(q0); }); };
    hooks.before_seams = [&] { if (att_thread.joinable()) att_thread.join(); };
  } else {
    build_att_tables();
  }
  auto c1 = tick();
  rc = run_edgebreaker(o.ct, o.eb, err, overlap ? &hooks : nullptr);
  if (flag_thread.joinable()) flag_thread.join();
  if (att_thread.joinable()) att_thread.join();
  if (seq_thread.joinable()) seq_thread.join();
  if (rc) return fail(rc, err);
  t_eb = since(c1);
  auto c2 = tick();
  if (!overlap) sequence_universal();
  ByteSink s;
  for (char ch : {'D', 'R', 'A', 'C', 'O'}) s.u8((uint8_t)ch);   // encode/header/mod.rs:26-54
  s.u8(2); s.u8(2); s.u8(1); s.u8(1); s.u16(0);
  s.bytes(o.eb.connectivity);
  bytes.swap(s.b);
  // views: attribute i uses the universal table when i == 0 or no attribute table i-1 exists
  // (all_inclusive_corner_table.rs:31-45)
  {
    std::vector<std::thread> th;   // sequences of attribute tables with seams: independent walks
    for (uint32_t i = 0; i < mesh->num_atts; ++i) {
      dmi_corner_table& v = o.views[i];
      v.num_faces = o.ct.F;
      v.corner_to_point = o.ct.c2p;
      const bool use_att = i > 0 && (i - 1) < o.ct.att.size();
      const bool seamless = !use_att || !o.ct.att[i - 1].interior_seams;
      if (use_att && !seamless) {
        const AttTable& t = o.ct.att[i - 1];
        v.num_vertices = t.num_vertices; v.corner_to_vertex = t.c2v.data(); v.opposite = t.opp.data(); v.left_most_corner = t.lmc.data();
        auto walk = [&o, i, &t] { TableRef tr{o.ct.F, t.num_vertices, t.c2v.data(), t.opp.data(), t.lmc.data()}; attribute_sequence(tr, o.eb, o.seqs[i]); };
        if (overlap) th.emplace_back(walk); else walk();
      } else {
        // a seam-free attribute table is identical to the universal one (same ids, same order)
        universal_view(v);
      }
    }
    for (auto& x : th) x.join();
    for (uint32_t i = 0; i < mesh->num_atts; ++i) {
      dmi_corner_table& v = o.views[i];
      const std::vector<uint32_t>& q = (v.corner_to_vertex == o.ct.c2v && i > 0) ? o.seqs[0] : o.seqs[i];   // (a table that IS the universal one shares its sequence)
      v.sequence = q.data();
      v.sequence_len = (uint32_t)q.size();
    }
  }
  if (!overlap) t_seq = since(c2);
  if (trace) { g_conn_us[0] += (uint64_t)(t_att * 1e3); g_conn_us[1] += (uint64_t)(t_eb * 1e3); g_conn_us[2] += (uint64_t)(t_seq * 1e3); g_conn_us[3] += (uint64_t)((since(c2) - (overlap ? 0 : t_seq)) * 1e3); }
  if (trace && mesh->num_faces > 100000) std::fprintf(stderr, "[dmi] host connectivity of %u faces (%s): universal corner table %.1f ms, attribute tables %.1f, Edgebreaker %.1f, universal sequencer %.1f, seam-table sequencers + views %.1f; total %.1f\n",
                          mesh->num_faces, overlap ? "overlapped: attribute tables and sequencer beside the Edgebreaker walk" : "in sequence", t_univ, t_att, t_eb, t_seq, since(c2), since(c0));
  return DMI_OK;
}

int dmi_init(int device, size_t staging_bytes, size_t device_bytes) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(DMI_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  NumaPin pin(device);   // (the staging is first touched — placed — on the GPU's memory node)
  auto holder = thread_stream(device);
  if (!holder) return fail(DMI_ERR_HIP, "hipStreamCreate");
  hipStream_t s = holder->s;
  {   // one launch per kernel file loads its code object: a 1-element prefix sum (dmi_conn.hip), a 4-word fill (dmi_relabel.hip), and the two below
    TempDev tmp;
    tmp.init(device, s, (size_t)1 << 20);
    uint32_t* w = tmp.take<uint32_t>(64);
    if (!w) return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc");
    HIP_TRY(hipMemsetAsync(w, 0, 256, s));
    launch_exclusive_scan_u32(w, 1, w + 8, s);
    launch_fill_u32(w + 16, 4, 0u, s);
    launch_last_corners(w, 1, w + 24, s);                                    // dmi_kernels.hip (corner 0 → point w[0] = 0)
    const CopyItem none{w + 32, 0, 0};
    HIP_TRY(hipMemcpyAsync(w + 40, &none, sizeof none, hipMemcpyHostToDevice, s));
    launch_scatter_items(reinterpret_cast<const CopyItem*>(w + 40), 1, reinterpret_cast<const uint8_t*>(w), s);   // dmi_chains.hip (an empty piece)
    HIP_TRY(hipStreamSynchronize(s));
  }
  if (staging_bytes) { HostStage* st = acquire_stage(device, staging_bytes); if (!st) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (staging)"); std::memset(st->p, 0, st->cap); release_stage(st); }
  if (device_bytes) { TempDev tmp; tmp.init(device, s, device_bytes); if (!tmp.take<uint8_t>(device_bytes)) return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc"); }
  return DMI_OK;
}

// The device half of the connectivity stage on its own (tests hold it against the host builders; dmi_mesh_prepare uses it internally).
int dmi_device_corner_table(const dmi_mesh* mesh, const dmi_config* cfg, uint32_t* opposite, uint32_t* left_most_corner, uint8_t* on_boundary, uint32_t* num_vertices, uint32_t* flags) {
  if (!mesh || !mesh->atts || mesh->num_atts == 0 || (!mesh->faces && mesh->num_faces) || !opposite || !num_vertices || !flags) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  DeviceTables dt;
  dt.device = cfg ? cfg->device : 0;
  auto holder = thread_stream(dt.device);
  if (!holder) return fail(DMI_ERR_HIP, "hipStreamCreate");
  dt.stream = cfg && cfg->stream ? static_cast<hipStream_t>(cfg->stream) : holder->s;
  std::vector<uint32_t> c2v_store;
  *flags = 0; *num_vertices = 0;
  const int rc = dt.build(mesh, c2v_store);
  *flags = dt.flags;
  if (rc) return rc;
  if (!dt.valid) return DMI_OK;   // flags say why: the host builder's case
  *num_vertices = dt.V;
  std::memcpy(opposite, dt.pre.opp, (size_t)mesh->num_faces * 12);
  if (left_most_corner) std::memcpy(left_most_corner, dt.pre.lmc, (size_t)dt.V * 4);
  if (on_boundary) std::memcpy(on_boundary, dt.pre.on_boundary, dt.V);
  return DMI_OK;
}

// One attribute corner table of ONE mesh built by the device kernels (k_att_* behind the universal table's kernels), read back — what the batch
// prepare runs for every (mesh, attribute) whose map differs from the position map, exposed so that tests hold it against the host builder.
int dmi_device_attribute_table(const dmi_mesh* mesh, const dmi_config* cfg, uint32_t att_index, uint8_t* seam_edge, uint32_t* corner_to_vertex, uint32_t* opposite,
                               uint32_t* left_most_corner, uint32_t* num_vertices, uint32_t* interior_seams, uint32_t* flags) {
  if (!mesh || !mesh->atts || att_index == 0 || att_index >= mesh->num_atts || !mesh->faces || !mesh->num_faces || !seam_edge || !corner_to_vertex || !opposite || !left_most_corner || !num_vertices || !interior_seams || !flags)
    return fail(DMI_ERR_INVALID_ARGUMENT, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  const int device = cfg ? cfg->device : 0;
  HIP_TRY(hipSetDevice(device));
  auto holder = thread_stream(device);
  if (!holder) return fail(DMI_ERR_HIP, "hipStreamCreate");
  hipStream_t s = holder->s;
  const dmi_attribute& pos = mesh->atts[0];
  const dmi_attribute& at = mesh->atts[att_index];
  const uint32_t F = mesh->num_faces, Vcap = pos.num_unique, P = pos.num_points;
  const size_t C = (size_t)F * 3, nv = (size_t)Vcap + 1, parts = scan_partials_words((uint32_t)nv);
  if (at.num_points < P) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute has fewer points than the Position attribute");
  AttStage st;
  const size_t map_pos_at = align256(C * 4), map_att_at = map_pos_at + (pos.point_to_value ? align256((size_t)P * 4) : 0), up = map_att_at + (at.point_to_value ? align256((size_t)at.num_points * 4) : 0);
  st.add(0, 0, F, Vcap, at.point_to_value ? (uint32_t)(map_att_at / 4) : kNone);
  const size_t rb_words = align256(up), host_need = st.layout(rb_words + 256 + align256(sizeof(ConnMeshDesc)));
  TempDev mem;
  mem.init(device, s, up + C * 4 * 4 + C + nv * 4 * 4 + nv + parts * 4 + st.device_bytes() + ((size_t)1 << 20));
  struct StageGuard { HostStage* st = nullptr; ~StageGuard() { release_stage(st); } } stage;
  stage.st = acquire_stage(device, host_need);
  if (!stage.st) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc");
  uint8_t* hp = stage.st->p;
  uint8_t* d_up = mem.take<uint8_t>(up);
  ConnArgs a{};
  uint32_t* d_c2v = pos.point_to_value ? mem.take<uint32_t>(C) : reinterpret_cast<uint32_t*>(d_up);
  a.opp = mem.take<uint32_t>(C); a.lmc = mem.take<uint32_t>(nv); a.on_boundary = mem.take<uint8_t>(nv);
  uint32_t* d_words = mem.take<uint32_t>(2);
  ConnMeshDesc* d_desc = mem.take<ConnMeshDesc>(1);
  a.ecount = mem.take<uint32_t>(nv); a.efill = mem.take<uint32_t>(nv); a.first = mem.take<uint32_t>(nv);
  a.he_key = mem.take<uint32_t>(C); a.he_corner = mem.take<uint32_t>(C); a.cdone = mem.take<uint8_t>(C); a.scan_partials = mem.take<uint32_t>(parts);
  if (!d_up || !d_c2v || !a.opp || !a.lmc || !a.on_boundary || !d_words || !d_desc || !a.ecount || !a.efill || !a.first || !a.he_key || !a.he_corner || !a.cdone || !a.scan_partials) return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc");
  std::memcpy(hp, mesh->faces, C * 4);
  if (pos.point_to_value) std::memcpy(hp + map_pos_at, pos.point_to_value, (size_t)P * 4);
  if (at.point_to_value) std::memcpy(hp + map_att_at, at.point_to_value, (size_t)at.num_points * 4);
  ConnMeshDesc* h_desc = reinterpret_cast<ConnMeshDesc*>(hp + rb_words + 256);
  *h_desc = ConnMeshDesc{0u, 0u, F, Vcap, pos.point_to_value ? (uint32_t)(map_pos_at / 4) : kNone, P, 0u, 0u};
  HIP_TRY(hipMemcpyAsync(d_up, hp, up, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_desc, h_desc, sizeof(ConnMeshDesc), hipMemcpyHostToDevice, s));
  a.meshes = d_desc; a.M = 1; a.total_faces = F; a.total_verts = Vcap;
  a.faces = reinterpret_cast<const uint32_t*>(d_up); a.p2v = reinterpret_cast<const uint32_t*>(d_up); a.c2v = d_c2v; a.flags = d_words; a.vmax = d_words + 1;
  HIP_TRY(conn_tables_clear(a, s));
  launch_conn_tables(a, s);
  HIP_TRY(hipMemcpyAsync(hp + rb_words, d_words, 8, hipMemcpyDeviceToHost, s));
  int rc = st.issue(a, mem, hp, s);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  *flags = reinterpret_cast<const uint32_t*>(hp + rb_words)[0];
  const AttInfo info = *reinterpret_cast<const AttInfo*>(hp + st.rb_info);
  *num_vertices = 0; *interior_seams = 0;
  if (!info.done) return DMI_OK;   // flags say why: the host builder's case
  *num_vertices = info.num_vertices; *interior_seams = info.interior;
  std::memcpy(seam_edge, hp + st.rb_seam, C);
  std::memcpy(corner_to_vertex, hp + st.rb_c2v, C * 4);
  std::memcpy(opposite, hp + st.rb_opp, C * 4);
  std::memcpy(left_most_corner, reinterpret_cast<const uint32_t*>(hp + st.rb_lmc) + info.pad, (size_t)info.num_vertices * 4);
  return DMI_OK;
}

int dmi_encode_connectivity(const dmi_mesh* mesh, dmi_buffer* header_and_connectivity, dmi_conn* conn) {
  if (!header_and_connectivity || !conn) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  auto* o = new ConnOwner();
  std::vector<uint8_t> bytes;
  int rc = build_connectivity(mesh, *o, bytes);
  if (rc) { delete o; return rc; }
  rc = to_buffer(bytes, header_and_connectivity);
  if (rc) { delete o; return rc; }
  conn->num_tables = (uint32_t)o->views.size();
  conn->tables = o->views.data();
  o->eb.materialize_seeds();
  conn->seeds = o->eb.seeds.data();
  conn->num_seeds = (uint32_t)o->eb.seeds.size();
  conn->owner = o;
  return DMI_OK;
}
void dmi_conn_free(dmi_conn* conn) {
  if (!conn) return;
  delete static_cast<ConnOwner*>(conn->owner);
  conn->owner = nullptr; conn->tables = nullptr; conn->seeds = nullptr; conn->num_tables = conn->num_seeds = 0;
}

// src (nullable): dmi_encode_mesh_device — the device copies of the faces / position map; mesh->atts[i].values are device pointers.
// Returns kNeedHostValues when such a mesh has to take the host builders (its values must come down first).
struct DeviceMeshSrc { const uint32_t* faces; const uint32_t* pos_map; };
static thread_local double g_tables_ms = 0;   // universal-table time of the running mesh_prepare_impl (device form)
constexpr int kNeedHostValues = -77;
static int mesh_prepare_impl(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** job, const DeviceMeshSrc* src);
int dmi_mesh_prepare(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** job) {
  return mesh_prepare_impl(mesh, cfg, header_and_connectivity, job, nullptr);
}
static int mesh_prepare_impl(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** job, const DeviceMeshSrc* src) {
  if (!header_and_connectivity || !job) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  g_tables_ms = 0;
  NumaPin pin(cfg ? cfg->device : 0);
  const bool trace = std::getenv("DMI_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  double t_conn = 0, t_create = 0, t_buf = 0;
  int rc;
  {
    std::unique_ptr<ConnOwner> op(new ConnOwner());
    ConnOwner& o = *op;
    std::vector<uint8_t> bytes;
    // The order-free half of the connectivity stage — universal corner table, left-most corners, boundary flags — runs on the device
    // for a mesh large enough to pay for the launches; its device copies then feed job creation directly.  (One of many meshes prepared
    // by dmi_meshes_prepare's workers takes this path only when it is large: per-mesh launches from many threads serialise in the runtime.)
    DeviceTables dt;
    struct Adopt { bool set = false; ~Adopt() { if (set) g_adopt_stream.reset(); } } adopt;
    const bool in_batch = (bool)g_adopt_stream;
    int ndev = 0;
    const bool want_device = mesh && mesh->atts && mesh->num_atts && mesh->faces && mesh->atts[0].att_type == DMI_ATT_POSITION &&
                             (src || (!std::getenv("DMI_HOST_CONNECTIVITY") && mesh->num_faces >= (in_batch ? kDeviceRelabelMinFaces : kDeviceTablesMinFaces))) &&
                             hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    if (want_device) {
      dt.device = cfg ? cfg->device : 0;
      if (cfg && cfg->stream) dt.stream = static_cast<hipStream_t>(cfg->stream);
      else {
        if (!g_adopt_stream) { g_adopt_stream = thread_stream(dt.device); adopt.set = (bool)g_adopt_stream; }   // the job created below shares the stream
        if (g_adopt_stream) dt.stream = g_adopt_stream->s;
      }
    }
    if (want_device && dt.stream) {
      if ((rc = dt.build(mesh, o.ct.c2v_own, src ? src->faces : nullptr, src ? src->pos_map : nullptr))) return rc;
      g_tables_ms = dt.t_down;
      if (trace) std::fprintf(stderr, "[dmi]   universal table of %u faces on the device: uploads issued %.2f ms, kernels + read-back issued %.2f, arrived %.2f (flags %#x)\n", mesh->num_faces, dt.t_up, dt.t_kernels, dt.t_down, dt.flags);
    }
    if (g_faces_event) { HIP_TRY(hipEventSynchronize(g_faces_event)); g_faces_event = nullptr; }   // (the host shadow of the faces / maps: in flight since before the table kernels)
    if (src && !dt.valid) return kNeedHostValues;
    rc = build_connectivity(mesh, o, bytes, dt.valid ? &dt.pre : nullptr, /*view_faces=*/true);
    if (rc) return rc;
    t_conn = ms();
    const DeviceTableView view{dt.d_faces, dt.d_c2v, dt.d_opp, true, src != nullptr};
    rc = job_create_impl(mesh->atts, o.views.data(), mesh->num_atts, nullptr, 0, cfg, dt.valid ? &view : nullptr, job);   // (every view carries its sequence: no seeds needed)
    if (rc) return rc;
    t_create = ms();
    rc = to_buffer(bytes, header_and_connectivity);
    if (rc) { dmi_job_destroy(*job); *job = nullptr; }
    t_buf = ms();
  }
  g_last_call = dmi_timings{};
  g_last_call.tables_ms = (float)g_tables_ms; g_last_call.connectivity_ms = (float)t_conn; g_last_call.job_create_ms = (float)(t_create - t_conn);
  if (!rc && *job) g_last_call.job_create_device_ms = (*job)->create_device_ms;
  if (trace && mesh->num_faces > 100000) std::fprintf(stderr, "[dmi] mesh_prepare: connectivity %.1f ms, job create %.1f, output buffer %.1f, release of the host tables %.1f\n", t_conn, t_create - t_conn, t_buf - t_create, ms() - t_buf);
  return rc;
}

// dmi_mesh_prepare for n independent meshes: the serial graph walks (corner tables, Edgebreaker, sequencers) and the
// uploads of different meshes run on a pool of host threads — the connectivity stage is the end-to-end bottleneck of a
// batch transcode once the attribute section is coded on the GPU.
static int meshes_prepare_impl(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, const int32_t* device_of_mesh, dmi_buffer* header_and_connectivity, dmi_job** jobs);
int dmi_meshes_prepare(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** jobs) {
  return meshes_prepare_impl(meshes, n, cfg, nullptr, header_and_connectivity, jobs);
}
// One process, several GPUs: mesh j is prepared on HIP device device_of_mesh[j] (dmi_shard_meshes deals them by triangle count).
int dmi_meshes_prepare_devices(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, const int32_t* device_of_mesh, dmi_buffer* header_and_connectivity, dmi_job** jobs) {
  if (!device_of_mesh) return fail(DMI_ERR_INVALID_ARGUMENT, "device_of_mesh is null");
  if (cfg && cfg->stream) return fail(DMI_ERR_INVALID_ARGUMENT, "a caller stream belongs to one device: leave dmi_config.stream null for a multi-device batch");
  const int ndev = dmi_device_count();
  for (uint32_t j = 0; j < n; ++j) if (device_of_mesh[j] < 0 || device_of_mesh[j] >= ndev) return fail(ndev ? DMI_ERR_INVALID_ARGUMENT : DMI_ERR_NO_DEVICE, "device ordinal out of range");
  return meshes_prepare_impl(meshes, n, cfg, device_of_mesh, header_and_connectivity, jobs);
}
// Greedy longest-processing-time deal of n meshes over n_devices by triangle count (the partition the multi-process form uses:
// draco-oxide_amd/distributed.py shard_indices): heaviest mesh first, each to the least loaded device, ties to the lower index.
int dmi_shard_meshes(const dmi_mesh* meshes, uint32_t n, uint32_t n_devices, int32_t* device_of_mesh) {
  if (!meshes || !device_of_mesh || n_devices == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  std::vector<uint32_t> order(n);
  for (uint32_t j = 0; j < n; ++j) order[j] = j;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return meshes[x].num_faces > meshes[y].num_faces; });
  std::vector<uint64_t> load(n_devices, 0);
  for (uint32_t j : order) {
    uint32_t best = 0;
    for (uint32_t d = 1; d < n_devices; ++d) if (load[d] < load[best]) best = d;
    device_of_mesh[j] = (int32_t)best;
    load[best] += meshes[j].num_faces;
  }
  return DMI_OK;
}

// ---- dmi_meshes_prepare, device form -------------------------------------------------------------------------------------------
// The meshes of a batch that live on one device go through the connectivity stage TOGETHER: their faces (and position maps) are packed
// into one staging copy, the universal corner tables of all of them come out of one launch per kernel (dmi_conn.hip) and back in one
// read-back; host threads then run what is serial per mesh (attribute tables, Edgebreaker, sequencers) and lay every job out in its
// own device memory without issuing device work; the coordinator finally uploads all sequences in one copy and runs the coding-order
// relabelling, the fan rows and the map compositions of all jobs in one launch per kernel.  Raw attribute values travel up beside the
// host walks.  A mesh the order-free table construction does not cover (device flags), or one with an attribute table of its own
// (interior seams), takes the per-mesh path (dmi_mesh_prepare) — same bytes either way (tests/test_gpu_batch_prepare.py).

// Two more library streams per (host thread, device): consecutive groups of a slice alternate between them, so the read-back of one
// group's tables overlaps the upload of the next group's faces (the two directions of the link run side by side).
static hipStream_t group_stream(int device, int which) {
  const std::shared_ptr<StreamHolder> h = thread_streams().get(1 + (which & 1), device);
  return h ? h->s : nullptr;
}

extern "C++" {
namespace dmi {
std::shared_ptr<StreamHolder> library_thread_stream(int device) { return thread_stream(device); }
hipStream_t library_group_stream(int device, int which) { return group_stream(device, which); }
NumaScope::NumaScope(int device) : impl(new NumaPin(device)) {}
NumaScope::~NumaScope() { delete static_cast<NumaPin*>(impl); }
}  // namespace dmi
}  // extern "C++"

namespace {
// One group of a slice: its meshes' faces / maps / values concatenated in one upload region, its tables in one read-back.
struct PrepGroup {
  struct MeshLay { size_t faces = 0, pos_map = (size_t)-1; std::vector<size_t> values, maps; uint32_t face_off = 0, vert_off = 0, Vcap = 0, desc_index = 0; bool mapped = false; };
  std::vector<uint32_t> which;   // mesh indices (into the caller's array)
  std::vector<MeshLay> lay;
  uint64_t total_faces = 0, total_verts = 0;
  uint32_t n_desc = 0;                 // connectivity descriptors of the group (= its meshes, or every member of an adopted built group)
  BuiltGroup* adopted = nullptr;       // the group is a device-built one (dmi_meshes_build): nothing to pack or upload
  AttStage att;                        // attribute corner tables the device builds for the group's meshes (host-packed groups; an adopted group's are its build's)
  size_t up_a = 0, up_b = 0, C = 0;
  bool any_mapped = false;
  hipStream_t S = nullptr;
  TempDev mem;
  HostStage* stage = nullptr;
  uint8_t* hp = nullptr;
  uint8_t* d_up = nullptr;
  const uint32_t* d_faces = nullptr;
  uint32_t *d_c2v = nullptr, *d_opp = nullptr;
  size_t rb_opp = 0, rb_c2v = 0, rb_lmc = 0, rb_onb = 0, rb_words = 0;
  hipEvent_t ev_tables = nullptr, ev_values = nullptr;
  hipEvent_t ev_tables_borrowed = nullptr;   // an adopted group's tables were issued by its build: the event belongs to the BuiltGroup
  bool tables_in = false;
  std::mutex wait_mutex;
  std::atomic<int> issued{0};   // 1: phase 1 of this group is through (its fields are final, its tables on their way); -1: phase 1 failed — the walkers give up
  ~PrepGroup() {
    if (S) (void)hipStreamSynchronize(S);
    if (ev_tables) (void)hipEventDestroy(ev_tables);
    if (ev_values) (void)hipEventDestroy(ev_values);
    release_stage(stage);
  }
  int wait_tables() {   // (any worker: the first one blocks on the event, the others on the mutex)
    std::lock_guard<std::mutex> lock(wait_mutex);
    if (tables_in) return DMI_OK;
    HIP_TRY(hipEventSynchronize(ev_tables_borrowed ? ev_tables_borrowed : ev_tables));
    tables_in = true;
    return DMI_OK;
  }
};
}  // namespace

extern "C++" {
void dmi::AttStage::add(uint32_t member, uint32_t k, uint32_t F, uint32_t vcap, uint32_t map_off_words) {
  items.push_back({member, k, (uint32_t)corners, (uint32_t)verts, F});
  descs.push_back(AttItemDesc{member, map_off_words, (uint32_t)corners, (uint32_t)verts});
  corners += (size_t)F * 3; verts += vcap;
}
size_t dmi::AttStage::layout(size_t at) {
  if (corners >= (1ull << 32) || verts >= (1ull << 32)) { items.clear(); descs.clear(); corners = verts = 0; }   // (too large for one launch: the host builds them)
  rb_items = at; rb_info = rb_items + align256(items.size() * sizeof(AttItemDesc)); rb_seam = rb_info + align256(items.size() * sizeof(AttInfo));
  rb_c2v = rb_seam + align256(corners); rb_opp = rb_c2v + align256(corners * 4); rb_lmc = rb_opp + align256(corners * 4);
  return rb_lmc + align256(corners * 4);
}
size_t dmi::AttStage::device_bytes() const { return corners * 13 + (verts + 1) * 5 + items.size() * (sizeof(AttItemDesc) + sizeof(AttInfo)) + scan_partials_words((uint32_t)verts + 1) * 4 + 4096; }
int dmi::AttStage::issue(const ConnArgs& a, TempDev& mem, uint8_t* host, hipStream_t s) {
  hp = host;
  if (items.empty()) return DMI_OK;
  AttArgs t{};
  t.n_items = (uint32_t)items.size(); t.total_corners = (uint32_t)corners; t.total_verts = (uint32_t)verts;
  AttItemDesc* d_items = mem.take<AttItemDesc>(items.size());
  t.seam = mem.take<uint8_t>(corners); t.vseam = mem.take<uint8_t>(verts); t.count = mem.take<uint32_t>(verts + 1);
  t.c2v = mem.take<uint32_t>(corners); t.opp = mem.take<uint32_t>(corners); t.lmc = mem.take<uint32_t>(corners);
  t.info = mem.take<AttInfo>(items.size()); t.scan_partials = mem.take<uint32_t>(scan_partials_words((uint32_t)verts + 1));
  if (!d_items || !t.seam || !t.vseam || !t.count || !t.c2v || !t.opp || !t.lmc || !t.info || !t.scan_partials) return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (attribute corner tables)");
  std::memcpy(host + rb_items, descs.data(), descs.size() * sizeof(AttItemDesc));
  HIP_TRY(hipMemcpyAsync(d_items, host + rb_items, items.size() * sizeof(AttItemDesc), hipMemcpyHostToDevice, s));
  t.items = d_items;
  HIP_TRY(att_tables_clear(t, s));
  launch_att_tables(a, t, s);
  d_c2v = t.c2v; d_opp = t.opp;
  HIP_TRY(hipMemcpyAsync(host + rb_info, t.info, items.size() * sizeof(AttInfo), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(host + rb_seam, t.seam, corners, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(host + rb_c2v, t.c2v, corners * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(host + rb_opp, t.opp, corners * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(host + rb_lmc, t.lmc, corners * 4, hipMemcpyDeviceToHost, s));
  return DMI_OK;
}
}  // extern "C++"
// the attribute tables the device built for one mesh of a group → pre.att (by index among the mesh's non-position attributes)
static void att_stage_fill(const AttStage& st, uint32_t member, uint32_t n_nonpos, std::vector<PrebuiltTable::Att>& out) {
  out.assign(n_nonpos, PrebuiltTable::Att{});
  if (st.items.empty() || !st.hp) return;
  const AttInfo* info = reinterpret_cast<const AttInfo*>(st.hp + st.rb_info);
  auto lo = std::lower_bound(st.items.begin(), st.items.end(), member, [](const AttStage::Item& x, uint32_t mi) { return x.member < mi; });
  for (auto it = lo; it != st.items.end() && it->member == member; ++it) {
    const size_t q = (size_t)(it - st.items.begin());
    if (!info[q].done || it->k >= n_nonpos) continue;
    PrebuiltTable::Att& pa = out[it->k];
    pa.ready = true; pa.interior = info[q].interior != 0; pa.nv = info[q].num_vertices;
    pa.seam = st.hp + st.rb_seam + it->corner_off;
    pa.c2v = reinterpret_cast<const uint32_t*>(st.hp + st.rb_c2v) + it->corner_off;
    pa.opp = reinterpret_cast<const uint32_t*>(st.hp + st.rb_opp) + it->corner_off;
    pa.lmc = reinterpret_cast<const uint32_t*>(st.hp + st.rb_lmc) + info[q].pad;
    pa.d_c2v = st.d_c2v + it->corner_off; pa.d_opp = st.d_opp + it->corner_off;
  }
}

// The universal corner tables of every member of a device-built group: descriptors up, the dmi_conn.hip kernels, the tables back into the
// group's own staging.  Nothing here waits; bg.conn.ev fires when the read-back has arrived.
extern "C++" int dmi::built_group_issue_tables(BuiltGroup& bg, hipStream_t s) {
  BuiltGroup::Conn& cn = bg.conn;
  if (cn.issued) return DMI_OK;
  HIP_TRY(hipSetDevice(bg.device));
  const uint32_t ND = (uint32_t)bg.members.size();
  uint64_t verts = 0;
  cn.any_mapped = false;
  for (const auto& mem : bg.members) { verts += mem.atts.empty() ? 0u : mem.atts[0].n_unique; cn.any_mapped = cn.any_mapped || (!mem.atts.empty() && mem.atts[0].map_off != (size_t)-1); }
  if (verts >= (1ull << 31) || bg.total_faces >= (1ull << 30)) return fail(DMI_ERR_INVALID_ARGUMENT, "built group too large");
  cn.total_verts = verts; cn.n_desc = ND;
  const size_t C = (size_t)bg.total_faces * 3, nv = (size_t)verts + 1, parts = scan_partials_words((uint32_t)nv);
  size_t att_bytes = 0;   // (upper bound of the attribute-table arrays: every non-position attribute a candidate)
  for (const auto& mem : bg.members) if (mem.atts.size() > 1) att_bytes += (mem.atts.size() - 1) * ((size_t)mem.F * 3 * 13 + (size_t)(mem.atts[0].n_unique + 1) * 5 + 512);
  cn.mem.init(bg.device, s, C * 4 * (cn.any_mapped ? 4 : 3) + C + nv * 4 * 4 + nv + parts * 4 + (size_t)ND * (sizeof(ConnMeshDesc) + 8) + att_bytes + ((size_t)2 << 20));
  const uint32_t* d_faces = reinterpret_cast<const uint32_t*>(bg.d_base);
  cn.d_c2v = cn.any_mapped ? cn.mem.take<uint32_t>(C) : const_cast<uint32_t*>(d_faces);
  cn.d_opp = cn.mem.take<uint32_t>(C);
  uint32_t* d_lmc = cn.mem.take<uint32_t>(nv);
  uint8_t* d_onb = cn.mem.take<uint8_t>(nv);
  uint32_t* d_words = cn.mem.take<uint32_t>((size_t)2 * ND);
  ConnMeshDesc* d_desc = cn.mem.take<ConnMeshDesc>(ND);
  ConnArgs a{};
  a.ecount = cn.mem.take<uint32_t>(nv); a.efill = cn.mem.take<uint32_t>(nv); a.first = cn.mem.take<uint32_t>(nv);
  a.he_key = cn.mem.take<uint32_t>(C); a.he_corner = cn.mem.take<uint32_t>(C);
  a.cdone = cn.mem.take<uint8_t>(C);
  a.scan_partials = cn.mem.take<uint32_t>(parts);
  if (!cn.d_c2v || !cn.d_opp || !d_lmc || !d_onb || !d_words || !d_desc || !a.ecount || !a.efill || !a.first || !a.he_key || !a.he_corner || !a.cdone || !a.scan_partials)
    return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (batch connectivity stage)");
  cn.rb_opp = 0; cn.rb_c2v = cn.rb_opp + align256(C * 4); cn.rb_lmc = cn.rb_c2v + (cn.any_mapped ? align256(C * 4) : 0); cn.rb_onb = cn.rb_lmc + align256(nv * 4);
  cn.rb_words = cn.rb_onb + align256(nv);
  // candidates for an attribute table of their own: non-position attributes whose map is not the position map entry for entry
  AttStage& st = cn.att;
  st = AttStage{};
  static const bool host_att = std::getenv("DMI_HOST_ATT_TABLES") != nullptr;
  for (uint32_t mi = 0; mi < ND && !host_att; ++mi) {
    const BuiltGroup::Member& mem = bg.members[mi];
    if (mem.atts.empty() || !mem.F) continue;
    uint32_t k = 0;
    for (size_t a = 0; a < mem.atts.size(); ++a) {
      if (mem.atts[a].att_type == DMI_ATT_POSITION) continue;
      const size_t ma = mem.atts[a].map_off, mp = mem.atts[0].map_off;
      const bool same = (ma == (size_t)-1 && mp == (size_t)-1) || (ma != (size_t)-1 && mp != (size_t)-1 && std::memcmp(bg.h_a + ma, bg.h_a + mp, (size_t)mem.P * 4) == 0);
      if (!same) st.add(mi, k, mem.F, mem.atts[0].n_unique, ma == (size_t)-1 ? kNone : (uint32_t)(ma / 4));
      ++k;
    }
  }
  const size_t rb_desc = cn.rb_words + align256((size_t)ND * 8);
  const size_t host_need = st.layout(rb_desc + align256((size_t)ND * sizeof(ConnMeshDesc)));
  cn.stage = acquire_stage(bg.device, host_need);
  if (!cn.stage) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (batch connectivity staging)");
  uint8_t* hp = cn.hp = cn.stage->p;
  ConnMeshDesc* h_desc = reinterpret_cast<ConnMeshDesc*>(hp + rb_desc);
  uint64_t vert = 0;
  for (uint32_t mi = 0; mi < ND; ++mi) {
    const BuiltGroup::Member& mem = bg.members[mi];
    const uint32_t vcap = mem.atts.empty() ? 0u : mem.atts[0].n_unique;
    const bool mapped = !mem.atts.empty() && mem.atts[0].map_off != (size_t)-1;
    h_desc[mi] = ConnMeshDesc{(uint32_t)(mem.faces_off / 12), (uint32_t)vert, mem.F, vcap, mapped ? (uint32_t)(mem.atts[0].map_off / 4) : kNone, mem.P, 0u, 0u};
    vert += vcap;
  }
  HIP_TRY(hipMemcpyAsync(d_desc, h_desc, (size_t)ND * sizeof(ConnMeshDesc), hipMemcpyHostToDevice, s));
  a.meshes = d_desc; a.M = ND; a.total_faces = (uint32_t)bg.total_faces; a.total_verts = (uint32_t)verts;
  a.faces = d_faces; a.p2v = reinterpret_cast<const uint32_t*>(bg.d_base); a.c2v = cn.d_c2v; a.opp = cn.d_opp; a.lmc = d_lmc; a.on_boundary = d_onb; a.flags = d_words; a.vmax = d_words + ND;
  int rc_att = DMI_OK;
  HIP_TRY(conn_tables_clear(a, s));
  launch_conn_tables(a, s);
  HIP_TRY(hipMemcpyAsync(hp + cn.rb_words, d_words, (size_t)ND * 8, hipMemcpyDeviceToHost, s));
  if (C) HIP_TRY(hipMemcpyAsync(hp + cn.rb_opp, cn.d_opp, C * 4, hipMemcpyDeviceToHost, s));
  if (cn.any_mapped && C) HIP_TRY(hipMemcpyAsync(hp + cn.rb_c2v, cn.d_c2v, C * 4, hipMemcpyDeviceToHost, s));
  if (verts) HIP_TRY(hipMemcpyAsync(hp + cn.rb_lmc, d_lmc, (size_t)verts * 4, hipMemcpyDeviceToHost, s));
  if (verts) HIP_TRY(hipMemcpyAsync(hp + cn.rb_onb, d_onb, (size_t)verts, hipMemcpyDeviceToHost, s));
  if ((rc_att = st.issue(a, cn.mem, hp, s))) return rc_att;
  HIP_TRY(hipEventCreateWithFlags(&cn.ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(cn.ev, s));
  cn.mem.pool.stream = nullptr; cn.mem.owner_waits = true;   // (the stream is a thread's library stream and the group may outlive the thread: its destructor waits for cn.ev instead)
  cn.issued = true;
  return DMI_OK;
}

// adopt (nullable): the groups are device-built ones (dmi_meshes_build) — which_all then lists their PRESENT members in group order
// (present[member] = index into the caller's arrays, -1 = not part of this call; the connectivity kernels run over every member: the
// arena's faces are one array)
struct AdoptedGroup { BuiltGroup* bg; std::vector<int32_t> present; };
static int prepare_slice_device(const dmi_mesh* meshes, const std::vector<uint32_t>& which_all, const dmi_config& cfg0, int device, uint32_t n_threads,
                                dmi_buffer* heads, dmi_job** jobs, std::vector<uint8_t>& done, const std::function<std::shared_ptr<StreamHolder>(uint32_t, int)>& worker_stream,
                                const std::vector<AdoptedGroup>* adopt = nullptr) {
  const uint32_t M = (uint32_t)which_all.size();
  if (!M) return DMI_OK;
  const bool trace = std::getenv("DMI_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  HIP_TRY(hipSetDevice(device));
  NumaPin pin(device);
  auto holder = thread_stream(device);
  if (!holder) return fail(DMI_ERR_HIP, "hipStreamCreate");
  hipStream_t S = holder->s;   // the coordinator's stream: job chunks are cleared on it, the deferred kernels of all jobs run on it
  struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } sync_on_exit{S};   // (also on error paths: a job the caller then destroys must not have its chunk cleared late)
  // ---- groups of ≈ 8M faces: the tables of the first arrive while the last is still being sent ----
  static const uint64_t group_faces = std::getenv("DMI_PREP_GROUP_FACES") ? (uint64_t)std::atoll(std::getenv("DMI_PREP_GROUP_FACES")) : (uint64_t)(3u << 20);   // (measured, 256 meshes / 11M faces: 6M 21.5–22 ms, 3M 19.4–20.4, 1.5M 22.6–23.1)
  std::vector<std::unique_ptr<PrepGroup>> groups;
  std::vector<std::pair<uint32_t, uint32_t>> where(M);   // position in which_all → (group, index within the group)
  if (adopt) {
    uint32_t k = 0;
    for (const AdoptedGroup& ag : *adopt) {
      groups.emplace_back(new PrepGroup());
      PrepGroup& g = *groups.back();
      g.adopted = ag.bg;
      uint64_t vert = 0;
      for (uint32_t mi = 0; mi < ag.bg->members.size(); ++mi) {
        const BuiltGroup::Member& mem = ag.bg->members[mi];
        const uint32_t vcap = mem.atts.empty() ? 0u : mem.atts[0].n_unique;
        const bool mapped = !mem.atts.empty() && mem.atts[0].map_off != (size_t)-1;
        g.any_mapped = g.any_mapped || mapped;
        if (ag.present[mi] >= 0) {
          if (k >= M || which_all[k] != (uint32_t)ag.present[mi]) return fail(DMI_ERR_INVALID_ARGUMENT, "adopted groups: member order");
          PrepGroup::MeshLay l;
          l.face_off = (uint32_t)(mem.faces_off / 12); l.vert_off = (uint32_t)vert; l.Vcap = vcap; l.desc_index = mi; l.mapped = mapped;
          where[k] = {(uint32_t)groups.size() - 1, (uint32_t)g.which.size()};
          g.which.push_back(which_all[k]);
          g.lay.push_back(std::move(l));
          ++k;
        }
        vert += vcap;
      }
      if (vert >= (1ull << 31) || ag.bg->total_faces >= (1ull << 30)) return fail(DMI_ERR_INVALID_ARGUMENT, "built group too large");
      g.total_faces = ag.bg->total_faces; g.total_verts = vert; g.n_desc = (uint32_t)ag.bg->members.size();
    }
    if (k != M) return fail(DMI_ERR_INVALID_ARGUMENT, "adopted groups: member count");
  }
  // (host meshes are dealt into groups largest first: the longest walks start first — the walkers take group after group — and none is left for the end)
  std::vector<uint32_t> by_size(adopt ? 0 : M);
  for (uint32_t k = 0; k < (uint32_t)by_size.size(); ++k) by_size[k] = k;
  std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) { return meshes[which_all[x]].num_faces > meshes[which_all[y]].num_faces; });
  for (uint32_t kq = 0; kq < M && !adopt; ++kq) {
    const uint32_t k = by_size[kq];
    const dmi_mesh& m = meshes[which_all[k]];
    // (the first group is a third of the others: its tables — what the walkers wait for at the start of the call — arrive that much sooner)
    if (groups.empty() || groups.back()->total_faces + m.num_faces > (groups.size() == 1 ? group_faces / 3 : group_faces)) {
      if (groups.empty() || groups.back()->total_faces) groups.emplace_back(new PrepGroup());
    }
    PrepGroup& g = *groups.back();
    PrepGroup::MeshLay l;
    l.face_off = (uint32_t)g.total_faces; l.vert_off = (uint32_t)g.total_verts; l.Vcap = m.atts[0].num_unique;
    l.desc_index = (uint32_t)g.which.size();
    l.mapped = m.atts[0].point_to_value != nullptr;
    g.any_mapped = g.any_mapped || l.mapped;
    g.total_faces += m.num_faces; g.total_verts += l.Vcap;
    where[k] = {(uint32_t)groups.size() - 1, (uint32_t)g.which.size()};
    g.which.push_back(which_all[k]);
    g.lay.push_back(std::move(l));
  }
  auto parallel_over = [&](uint32_t count, const std::function<int(uint32_t, uint32_t)>& fn, const std::function<uint64_t(uint32_t)>& weight, uint32_t max_threads = 0) -> int {   // fn(worker, i), heaviest first unless weight is null
    std::vector<uint32_t> order(count);
    for (uint32_t k = 0; k < count; ++k) order[k] = k;
    if (weight) std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return weight(x) > weight(y); });
    std::atomic<uint32_t> next{0};
    const uint32_t nt = std::max(1u, std::min(max_threads ? std::min(max_threads, n_threads) : n_threads, count));
    std::vector<int> rcs(nt, DMI_OK);
    std::vector<std::string> errs(nt);
    auto work = [&](uint32_t t) {
      (void)hipSetDevice(device);
      for (uint32_t i; (i = next.fetch_add(1)) < count;) { const int rc = fn(t, order[i]); if (rc) { rcs[t] = rc; errs[t] = g_last_error; next.store(count); return; } }
    };
    if (nt == 1) work(0);
    else { std::vector<std::thread> th; for (uint32_t t = 0; t < nt; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
    for (uint32_t t = 0; t < nt; ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
    return DMI_OK;
  };
  int rc;
  // ---- phase 2 (set up here, started behind the first group's phase 1): host walks + job layout per mesh, group by group as their tables arrive ----
  std::vector<std::unique_ptr<ConnOwner>> owners(M);
  std::vector<JobDefer> defers(M);
  std::vector<uint8_t> deferred(M, 0);
  std::vector<uint32_t> walk_order(M);   // group order; inside a group the largest mesh first
  for (uint32_t k = 0; k < M; ++k) walk_order[k] = k;
  std::stable_sort(walk_order.begin(), walk_order.end(), [&](uint32_t x, uint32_t y) {
    if (where[x].first != where[y].first) return where[x].first < where[y].first;
    return meshes[which_all[x]].num_faces > meshes[which_all[y]].num_faces;
  });
  std::atomic<uint64_t> ns_wait{0}, ns_conn{0}, ns_job{0}, ns_buf{0};   // thread time by step (trace)
  auto now_ns = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const std::function<int(uint32_t, uint32_t)> walk_one = [&](uint32_t t, uint32_t i) -> int {
    const uint32_t kk = walk_order[i];
    struct Busy { bool was = g_batch_worker_busy; ~Busy() { g_batch_worker_busy = was; } } busy;
    g_batch_worker_busy = M >= 2 * n_threads && n_threads > 1;
    PrepGroup& g = *groups[where[kk].first];
    const uint32_t k = where[kk].second, j = g.which[k];
    const dmi_mesh& m = meshes[j];
    const uint64_t w0 = now_ns();
    for (int st; (st = g.issued.load(std::memory_order_acquire)) != 1;) {   // (the coordinator is still packing / sending this group)
      if (st < 0) return DMI_ERR_HIP;
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (!g.adopted) {   // this mesh's values → staging → device, behind the group's table kernels on its stream
      size_t lo = (size_t)-1, hi = 0;
      for (uint32_t a = 0; a < m.num_atts; ++a) {
        const dmi_attribute& at = m.atts[a];
        const size_t vb = (size_t)at.num_unique * at.num_components * 4;
        if (!vb) continue;
        std::memcpy(g.hp + g.lay[k].values[a], at.values, vb);
        lo = std::min(lo, g.lay[k].values[a]); hi = std::max(hi, g.lay[k].values[a] + vb);
      }
      if (hi > lo) HIP_TRY(hipMemcpyAsync(g.d_up + lo, g.hp + lo, hi - lo, hipMemcpyHostToDevice, g.S));
    }
    int r = g.wait_tables();
    if (r) return r;
    struct Slot { Slot() { walk_slots().acquire(); } ~Slot() { walk_slots().release(); } } slot;   // (the process-wide budget of running walks: dmi_host.hpp)
    const uint64_t w1 = now_ns();
    ns_wait += w1 - w0;
    auto bail = [&](int code, const std::string& what) { return fail(code, "mesh " + std::to_string(j) + ": " + what); };
    const uint8_t* hp = g.hp;
    const uint32_t* h_words = reinterpret_cast<const uint32_t*>(hp + g.rb_words);
    const uint32_t flags = h_words[g.lay[k].desc_index];
    if (flags & CONN_BAD_INDEX) return bail(DMI_ERR_INVALID_ARGUMENT, "face index ≥ number of points, or a position value index out of range");
    if (flags & (CONN_DEGENERATE | CONN_NONMANIFOLD_EDGE | CONN_MULTI_FAN)) return DMI_OK;   // the per-mesh path (the reference's serial walks)
    if (flags & CONN_UNUSED_VERTEX) return bail(DMI_ERR_UNUSED_VERTICES, "mesh contains unused vertices");
    PrebuiltTable pre;
    const size_t cb = (size_t)g.lay[k].face_off * 3;
    pre.c2v = g.lay[k].mapped ? reinterpret_cast<const uint32_t*>(hp + g.rb_c2v) + cb : m.faces;
    pre.opp = reinterpret_cast<const uint32_t*>(hp + g.rb_opp) + cb;
    pre.lmc = reinterpret_cast<const uint32_t*>(hp + g.rb_lmc) + g.lay[k].vert_off;
    pre.on_boundary = hp + g.rb_onb + g.lay[k].vert_off;
    pre.V = h_words[g.n_desc + g.lay[k].desc_index] + 1;
    pre.no_boundary = !(flags & CONN_HAS_BOUNDARY);
    {   // attribute tables the device built for this mesh (k_att_*)
      uint32_t n_nonpos = 0;
      for (uint32_t a = 0; a < m.num_atts; ++a) n_nonpos += m.atts[a].att_type != DMI_ATT_POSITION;
      att_stage_fill(g.adopted ? g.adopted->conn.att : g.att, g.lay[k].desc_index, n_nonpos, pre.att);
    }
    owners[kk].reset(new ConnOwner());
    ConnOwner& o = *owners[kk];
    std::vector<uint8_t> bytes;
    if ((r = build_connectivity(&m, o, bytes, &pre, /*view_faces=*/true))) return bail(r, g_last_error);
    const uint64_t w2 = now_ns();
    ns_conn += w2 - w1;
    dmi_config c = cfg0;
    c.device = device;
    g_adopt_stream = worker_stream(t % kPrepareStreams, device);   // the job's own stream for its encodes
    struct Drop { ~Drop() { g_adopt_stream.reset(); } } drop;
    // (a mesh with attribute tables of its own — interior seams — is deferred like the others: its seam tables go up with the sequences)
    static const bool defer_seams = !std::getenv("DMI_NO_DEFER_SEAMS");
    bool all_universal = true;
    for (uint32_t a = 1; a < m.num_atts; ++a) all_universal = all_universal && o.views[a].corner_to_vertex == o.views[0].corner_to_vertex && o.views[a].opposite == o.views[0].opposite;
    if (all_universal || defer_seams) {
      JobDefer& d = defers[kk];
      d.stream = S;
      d.values_dev.assign(m.num_atts, nullptr); d.maps_dev.assign(m.num_atts, nullptr);
      for (uint32_t a = 0; a < m.num_atts; ++a) {
        if (g.lay[k].values[a] != (size_t)-1) d.values_dev[a] = g.d_up + g.lay[k].values[a];
        if (g.lay[k].maps[a] != (size_t)-1) d.maps_dev[a] = reinterpret_cast<const uint32_t*>(g.d_up + g.lay[k].maps[a]);
      }
      DeviceTableView view{g.d_faces + cb, g.d_c2v + cb, g.d_opp + cb, true};
      // attribute tables the device built stay where they are: the batched relabelling reads them (key = the host copy the walks used)
      std::vector<const uint32_t*> att_key, att_c2v, att_opp;
      for (size_t q = 0; q < pre.att.size() && q < o.ct.att.size(); ++q)
        if (pre.att[q].ready && pre.att[q].interior && !o.ct.att[q].c2v.empty()) { att_key.push_back(o.ct.att[q].c2v.data()); att_c2v.push_back(pre.att[q].d_c2v); att_opp.push_back(pre.att[q].d_opp); }
      view.n_att = (uint32_t)att_key.size(); view.att_key = att_key.data(); view.att_c2v = att_c2v.data(); view.att_opp = att_opp.data();
      r = job_create_impl(m.atts, o.views.data(), m.num_atts, nullptr, 0, &c, &view, &jobs[j], &d);
      deferred[kk] = r == DMI_OK;
    } else if (g.adopted) {   // an attribute table of its own, values resident in the built group: the universal table from the device, the seam tables from the host
      std::vector<dmi_attribute> atts_dev(m.atts, m.atts + m.num_atts);
      for (uint32_t a = 0; a < m.num_atts; ++a) atts_dev[a].values = g.lay[k].values[a] != (size_t)-1 ? static_cast<const void*>(g.d_up + g.lay[k].values[a]) : nullptr;
      const DeviceTableView view{g.d_faces + cb, g.d_c2v + cb, g.d_opp + cb, true, true};
      r = job_create_impl(atts_dev.data(), o.views.data(), m.num_atts, nullptr, 0, &c, &view, &jobs[j], nullptr);
      owners[kk].reset();
    } else {   // an attribute table of its own: the host relabelling form reads the tables where the walks read them
      r = job_create_impl(m.atts, o.views.data(), m.num_atts, nullptr, 0, &c, nullptr, &jobs[j], nullptr);
      owners[kk].reset();
    }
    if (r) return bail(r, g_last_error);
    const uint64_t w3 = now_ns();
    ns_job += w3 - w2;
    if ((r = to_buffer(bytes, &heads[j]))) return r;
    ns_buf += now_ns() - w3;
    done[j] = 1;
    return DMI_OK;
  };
  // The walkers start as soon as the FIRST group is on its way: the packing and sending of the later groups (a few threads of their own) runs
  // beside the walks of the earlier ones (phase 1 used to finish for all groups first: 4–5 ms of a 20 ms prepare with every walker idle).
  std::thread walkers;
  int rc_walk = DMI_OK;
  std::string err_walk;
  struct JoinWalkers { std::thread& t; std::vector<std::unique_ptr<PrepGroup>>& gs; ~JoinWalkers() { if (t.joinable()) { for (auto& g : gs) { int z = 0; g->issued.compare_exchange_strong(z, -1); } t.join(); } } } join_walkers{walkers, groups};
  auto pack_threads = [&]() -> uint32_t { return walkers.joinable() ? std::max(2u, n_threads / 4) : 0u; };   // (0 = all: nothing else runs yet)
  // ---- phase 1, group by group: layout, pack, send, build the tables, fetch them (nothing here waits for the device) ----
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    PrepGroup& g = *groups[gi];
    const uint32_t Mg = (uint32_t)g.which.size();
    g.S = group_stream(device, (int)(gi & 1));
    if (!g.S) return fail(DMI_ERR_HIP, "hipStreamCreate");
    g.C = (size_t)g.total_faces * 3;
    if (g.adopted) {
      // a device-built group: faces (one array), maps and values are where dmi_meshes_build left them; offsets are bytes from its base
      BuiltGroup& bg = *g.adopted;
      bool want_lmc = false;
      for (uint32_t k = 0; k < Mg; ++k) {
        const dmi_mesh& m = meshes[g.which[k]];
        const BuiltGroup::Member& mem = bg.members[g.lay[k].desc_index];
        PrepGroup::MeshLay& l = g.lay[k];
        if (mem.atts.size() != m.num_atts) return fail(DMI_ERR_INVALID_ARGUMENT, "built mesh: attribute count");
        l.values.assign(m.num_atts, (size_t)-1); l.maps.assign(m.num_atts, (size_t)-1);
        for (uint32_t i = 0; i < m.num_atts; ++i) {
          if (mem.atts[i].n_unique) l.values[i] = mem.atts[i].val_off;
          l.maps[i] = mem.atts[i].map_off;
          if (m.atts[i].att_type != DMI_ATT_POSITION && m.atts[i].point_to_value != m.atts[0].point_to_value) want_lmc = true;
        }
        l.pos_map = l.maps[0];
        l.faces = mem.faces_off;
      }
      (void)want_lmc;   // (left-most corners always come back for a built group: 4 bytes per vertex)
      if (!bg.conn.issued && (rc = built_group_issue_tables(bg, g.S))) return rc;
      g.d_up = bg.d_base;
      g.d_faces = reinterpret_cast<const uint32_t*>(bg.d_base);
      g.d_c2v = bg.conn.d_c2v; g.d_opp = bg.conn.d_opp;
      g.hp = bg.conn.hp;
      g.rb_opp = bg.conn.rb_opp; g.rb_c2v = bg.conn.rb_c2v; g.rb_lmc = bg.conn.rb_lmc; g.rb_onb = bg.conn.rb_onb; g.rb_words = bg.conn.rb_words;
      g.ev_tables_borrowed = bg.conn.ev;
      g.issued.store(1, std::memory_order_release);
      if (!walkers.joinable() && M > 1) walkers = std::thread([&] { rc_walk = parallel_over(M, walk_one, nullptr); if (rc_walk) err_walk = g_last_error; });
      continue;
    }
    g.n_desc = Mg;
    {   // faces are ONE array (global corner index = 3·face_off + local corner): no per-mesh padding
      size_t at = 0;
      for (uint32_t k = 0; k < Mg; ++k) { g.lay[k].faces = at; at += (size_t)meshes[g.which[k]].num_faces * 12; }
      g.up_a = align256(at);
      for (uint32_t k = 0; k < Mg; ++k) if (g.lay[k].mapped) { g.lay[k].pos_map = g.up_a; g.up_a = align256(g.up_a + (size_t)meshes[g.which[k]].atts[0].num_points * 4); }
      // the other attributes' maps ride in part A too (the attribute-table kernels read them right behind the universal tables)
      for (uint32_t k = 0; k < Mg; ++k) {
        const dmi_mesh& m = meshes[g.which[k]];
        g.lay[k].maps.assign(m.num_atts, (size_t)-1);
        for (uint32_t i = 1; i < m.num_atts; ++i) {
          const dmi_attribute& a = m.atts[i];
          if (!a.point_to_value) continue;
          bool shared = a.point_to_value == m.atts[0].point_to_value;
          if (shared) g.lay[k].maps[i] = g.lay[k].pos_map;
          for (uint32_t j = 1; j < i && !shared; ++j) if (m.atts[j].point_to_value == a.point_to_value) { g.lay[k].maps[i] = g.lay[k].maps[j]; shared = true; }
          if (!shared) { g.lay[k].maps[i] = g.up_a; g.up_a = align256(g.up_a + (size_t)a.num_points * 4); }
        }
      }
    }
    bool want_lmc = false;   // left-most corners are only read by the host builder of attribute tables (an attribute indexed unlike the Position attribute)
    for (uint32_t k = 0; k < Mg; ++k) {
      const dmi_mesh& m = meshes[g.which[k]];
      PrepGroup::MeshLay& l = g.lay[k];
      l.values.assign(m.num_atts, (size_t)-1);
      for (uint32_t i = 0; i < m.num_atts; ++i) {
        const dmi_attribute& a = m.atts[i];
        const size_t vb = (size_t)a.num_unique * a.num_components * 4;
        if (vb) { l.values[i] = g.up_a + g.up_b; g.up_b = align256(g.up_b + ((vb + 15) & ~(size_t)15)); }
        if (a.att_type != DMI_ATT_POSITION && a.point_to_value != m.atts[0].point_to_value) want_lmc = true;
        if (a.point_to_value) {
          if (i == 0) l.maps[i] = l.pos_map;
          else for (uint32_t j = 0; j < i; ++j) if (m.atts[j].point_to_value == a.point_to_value) { l.maps[i] = l.maps[j]; break; }   // (placed in part A above)
        }
      }
    }
    // attribute corner tables on the device for the attributes whose maps are not the position map entry for entry
    {
      static const bool host_att = std::getenv("DMI_HOST_ATT_TABLES") != nullptr;
      for (uint32_t k = 0; k < Mg && !host_att; ++k) {
        const dmi_mesh& m = meshes[g.which[k]];
        if (!m.num_faces) continue;
        uint32_t idx = 0;
        for (uint32_t i = 0; i < m.num_atts; ++i) {
          const dmi_attribute& a = m.atts[i];
          if (a.att_type == DMI_ATT_POSITION) continue;
          const uint32_t* pm = m.atts[0].point_to_value;
          const bool same = a.point_to_value == pm || (a.point_to_value && pm && a.num_points == m.atts[0].num_points && std::memcmp(a.point_to_value, pm, (size_t)a.num_points * 4) == 0);
          bool earlier = false;   // (an attribute with the map array of an earlier one copies that one's table on the host)
          for (uint32_t j = 1; j < i && !earlier; ++j) earlier = m.atts[j].att_type != DMI_ATT_POSITION && m.atts[j].point_to_value == a.point_to_value && a.point_to_value;
          if (!same && !earlier) g.att.add(k, idx, m.num_faces, g.lay[k].Vcap, a.point_to_value ? (uint32_t)(g.lay[k].maps[i] / 4) : kNone);
          ++idx;
        }
      }
    }
    const size_t up_bytes = g.up_a + g.up_b, C = g.C;
    const size_t nv = (size_t)g.total_verts + 1, parts = scan_partials_words((uint32_t)nv);
    const size_t rb_desc0 = align256(up_bytes) + align256(C * 4) + (g.any_mapped ? align256(C * 4) : 0) + align256(nv * 4) + align256(nv) + align256((size_t)Mg * 8);
    const size_t host_need = g.att.layout(rb_desc0 + align256((size_t)Mg * sizeof(ConnMeshDesc)));   // (may drop the items: before the device memory is sized)
    g.mem.init(device, g.S, up_bytes + C * 4 * (g.any_mapped ? 4 : 3) + C + nv * 4 * 4 + nv + parts * 4 + (size_t)Mg * (sizeof(ConnMeshDesc) + 8) + g.att.device_bytes() + ((size_t)1 << 20));
    g.d_up = g.mem.take<uint8_t>(up_bytes);
    g.d_faces = reinterpret_cast<const uint32_t*>(g.d_up);
    g.d_c2v = g.any_mapped ? g.mem.take<uint32_t>(C) : const_cast<uint32_t*>(g.d_faces);
    g.d_opp = g.mem.take<uint32_t>(C);
    uint32_t* d_lmc = g.mem.take<uint32_t>(nv);
    uint8_t* d_onb = g.mem.take<uint8_t>(nv);
    uint32_t* d_words = g.mem.take<uint32_t>((size_t)2 * Mg);
    ConnMeshDesc* d_desc = g.mem.take<ConnMeshDesc>(Mg);
    ConnArgs a{};
    a.ecount = g.mem.take<uint32_t>(nv); a.efill = g.mem.take<uint32_t>(nv); a.first = g.mem.take<uint32_t>(nv);
    a.he_key = g.mem.take<uint32_t>(C); a.he_corner = g.mem.take<uint32_t>(C);
    a.cdone = g.mem.take<uint8_t>(C);
    a.scan_partials = g.mem.take<uint32_t>(parts);
    if (!g.d_up || !g.d_c2v || !g.d_opp || !d_lmc || !d_onb || !d_words || !d_desc || !a.ecount || !a.efill || !a.first || !a.he_key || !a.he_corner || !a.cdone || !a.scan_partials)
      return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (batch connectivity stage)");
    // host: staging of the upload | read-back: opp, [c2v], lmc, on_boundary, flags/vmax | descriptors
    g.rb_opp = align256(up_bytes); g.rb_c2v = g.rb_opp + align256(C * 4); g.rb_lmc = g.rb_c2v + (g.any_mapped ? align256(C * 4) : 0); g.rb_onb = g.rb_lmc + align256(nv * 4);
    g.rb_words = g.rb_onb + align256(nv);
    const size_t rb_desc = g.rb_words + align256((size_t)Mg * 8);
    if (rb_desc != rb_desc0) return fail(DMI_ERR_HIP, "batch connectivity staging layout");
    g.stage = acquire_stage(device, host_need);
    if (!g.stage) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (batch connectivity staging)");
    uint8_t* hp = g.hp = g.stage->p;
    ConnMeshDesc* h_desc = reinterpret_cast<ConnMeshDesc*>(hp + rb_desc);
    for (uint32_t k = 0; k < Mg; ++k) {
      const dmi_mesh& m = meshes[g.which[k]];
      h_desc[k] = ConnMeshDesc{g.lay[k].face_off, g.lay[k].vert_off, m.num_faces, g.lay[k].Vcap, g.lay[k].mapped ? (uint32_t)(g.lay[k].pos_map / 4) : kNone, m.atts[0].num_points, 0u, 0u};
    }
    auto faces_of = [&](uint32_t k) { return (uint64_t)meshes[g.which[k]].num_faces; };
    if ((rc = parallel_over(Mg, [&](uint32_t, uint32_t k) -> int {
          const dmi_mesh& m = meshes[g.which[k]];
          std::memcpy(hp + g.lay[k].faces, m.faces, (size_t)m.num_faces * 12);
          if (g.lay[k].mapped) std::memcpy(hp + g.lay[k].pos_map, m.atts[0].point_to_value, (size_t)m.atts[0].num_points * 4);
          for (uint32_t i = 1; i < m.num_atts; ++i) {
            const dmi_attribute& at = m.atts[i];
            if (!at.point_to_value || at.point_to_value == m.atts[0].point_to_value) continue;
            bool first = true;
            for (uint32_t j = 1; j < i; ++j) if (m.atts[j].point_to_value == at.point_to_value) first = false;
            if (first) std::memcpy(hp + g.lay[k].maps[i], at.point_to_value, (size_t)at.num_points * 4);
          }
          return DMI_OK;
        }, faces_of, pack_threads()))) return rc;
    HIP_TRY(hipMemcpyAsync(g.d_up, hp, g.up_a, hipMemcpyHostToDevice, g.S));
    HIP_TRY(hipMemcpyAsync(d_desc, h_desc, (size_t)Mg * sizeof(ConnMeshDesc), hipMemcpyHostToDevice, g.S));
    a.meshes = d_desc; a.M = Mg; a.total_faces = (uint32_t)g.total_faces; a.total_verts = (uint32_t)g.total_verts;
    a.faces = g.d_faces; a.p2v = reinterpret_cast<const uint32_t*>(g.d_up); a.c2v = g.d_c2v; a.opp = g.d_opp; a.lmc = d_lmc; a.on_boundary = d_onb; a.flags = d_words; a.vmax = d_words + Mg;
    HIP_TRY(conn_tables_clear(a, g.S));
    launch_conn_tables(a, g.S);
    HIP_TRY(hipMemcpyAsync(hp + g.rb_words, d_words, (size_t)Mg * 8, hipMemcpyDeviceToHost, g.S));
    HIP_TRY(hipMemcpyAsync(hp + g.rb_opp, g.d_opp, C * 4, hipMemcpyDeviceToHost, g.S));
    if (g.any_mapped) HIP_TRY(hipMemcpyAsync(hp + g.rb_c2v, g.d_c2v, C * 4, hipMemcpyDeviceToHost, g.S));
    if (want_lmc) HIP_TRY(hipMemcpyAsync(hp + g.rb_lmc, d_lmc, (size_t)g.total_verts * 4, hipMemcpyDeviceToHost, g.S));
    HIP_TRY(hipMemcpyAsync(hp + g.rb_onb, d_onb, (size_t)g.total_verts, hipMemcpyDeviceToHost, g.S));
    if (!g.att.items.empty()) { want_lmc = true; HIP_TRY(hipMemcpyAsync(hp + g.rb_lmc, d_lmc, (size_t)g.total_verts * 4, hipMemcpyDeviceToHost, g.S)); }
    if ((rc = g.att.issue(a, g.mem, hp, g.S))) return rc;
    HIP_TRY(hipEventCreateWithFlags(&g.ev_tables, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(g.ev_tables, g.S));
    // part B (the values — more than half of the bytes) is packed and sent mesh by mesh by the walkers, first thing, while they would otherwise
    // wait for this group's tables (walk_one): phase 1 — what every walker waits for — packs faces and maps only
    HIP_TRY(hipEventCreateWithFlags(&g.ev_values, hipEventDisableTiming));
    g.issued.store(1, std::memory_order_release);
    if (!walkers.joinable() && M > 1) walkers = std::thread([&] { rc_walk = parallel_over(M, walk_one, nullptr); if (rc_walk) err_walk = g_last_error; });
  }
  const double t_issue = ms();
  if (walkers.joinable()) { walkers.join(); rc = rc_walk; if (rc) fail(rc, err_walk); }
  else rc = parallel_over(M, walk_one, nullptr);   // (one mesh)
  for (auto& g : groups) if (!rc && g->ev_values) HIP_TRY(hipEventRecord(g->ev_values, g->S));   // (behind the last of the walkers' value copies)
  if (rc) return rc;   // (the groups' destructors wait for their streams)
  const double t_walks = ms();
  // ---- phase 3: all deferred device work: sequences up in one copy, then one launch per kernel on the coordinator's stream ----
  std::vector<RelabelItem> items;
  std::vector<FanItem> fans;
  std::vector<ComposeItem> comps;
  struct Move { void* dst; const void* src; size_t bytes; };
  std::vector<Move> moves;
  std::vector<CopyItem> clears;   // ranges of the jobs' uncleared chunks that must start as zeros (JobDefer::clears)
  std::vector<const uint32_t*> seq_src, c2v_src, opp_src;   // host sources per item (c2v / opp: attribute tables of their own)
  uint64_t rf = 0, rv = 0, rk = 0, rs = 0, rr = 0, fan_total = 0, comp_total = 0, host_table_words = 0;
  for (uint32_t kk = 0; kk < M; ++kk) {
    if (!deferred[kk]) continue;
    JobDefer& d = defers[kk];
    const uint32_t first_item = (uint32_t)items.size();
    for (size_t q = 0; q < d.relabels.size(); ++q) {
      RelabelItem it = d.relabels[q];
      const bool universal = q == 0;
      it.order_item = first_item;
      it.face_off = (uint32_t)rf; it.vert_off = (uint32_t)rv; it.key_off = (uint32_t)rk; it.seq_off = (uint32_t)rs; it.remap_off = (uint32_t)rr;
      seq_src.push_back(it.seq); c2v_src.push_back(d.host_c2v[q]); opp_src.push_back(d.host_opp[q]);
      if (universal) { rf += it.F; rk += (uint64_t)it.n_seq + 1; }
      else host_table_words += 6ull * it.F;
      rv += it.V; rs += it.n_seq; rr += it.F;
      items.push_back(it);
    }
    for (FanItem f : d.fans) { f.off = (uint32_t)fan_total; fan_total += f.n; fans.push_back(f); }
    for (ComposeItem ci : d.compose) { ci.off = (uint32_t)comp_total; comp_total += ci.n; comps.push_back(ci); }
    for (const auto& cp : d.copies) moves.push_back({cp.dst, cp.src_dev, (cp.bytes + 15) & ~(size_t)15});
    for (const auto& cl : d.clears) clears.push_back(CopyItem{cl.p, 0u, (uint64_t)cl.bytes});
  }
  if (rf >= (1ull << 32) / 3 || rr >= (1ull << 32) / 3 || rv >= (1ull << 32) || rk >= (1ull << 32) || fan_total >= (1ull << 32) || comp_total >= (1ull << 32) || (rs + host_table_words) * 4 >= (1ull << 36))
    return fail(DMI_ERR_INVALID_ARGUMENT, "batch slice too large");
  for (auto& g : groups) if (g->ev_values) HIP_TRY(hipStreamWaitEvent(S, g->ev_values, 0));   // the values the jobs copy from must have arrived (a built group's are: its build waited)
  for (auto& g : groups) if (g->ev_tables_borrowed) HIP_TRY(hipStreamWaitEvent(S, g->ev_tables_borrowed, 0));   // (its device tables, which the relabelling reads)
  TempDev mem3;
  struct StageGuard { HostStage* st = nullptr; ~StageGuard() { release_stage(st); } } stage3;
  if (!items.empty() || !moves.empty() || !clears.empty()) {
    // CopyItem{destination, offset of the source relative to `base`, bytes}: one base for all groups' regions
    const uint8_t* base = nullptr;
    for (auto& mv : moves) if (!base || static_cast<const uint8_t*>(mv.src) < base) base = static_cast<const uint8_t*>(mv.src);
    std::vector<CopyItem> copies;
    copies.reserve(moves.size());
    for (auto& mv : moves) copies.push_back(CopyItem{mv.dst, (uint64_t)(static_cast<const uint8_t*>(mv.src) - base), (uint64_t)mv.bytes});
    // staging: [sequences of every item | vertex ids + opposite corners of the attribute tables of their own | descriptors]
    const size_t off_tables = align256((size_t)rs * 4);
    const size_t off_items = off_tables + align256((size_t)host_table_words * 4), off_fans = off_items + align256(items.size() * sizeof(RelabelItem)), off_comps = off_fans + align256(fans.size() * sizeof(FanItem)),
                 off_copies = off_comps + align256(comps.size() * sizeof(ComposeItem)), off_clears = off_copies + align256(copies.size() * sizeof(CopyItem)),
                 need2 = off_clears + align256(clears.size() * sizeof(CopyItem));
    stage3.st = acquire_stage(device, need2);
    if (!stage3.st) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (batch sequences staging)");
    uint8_t* h2 = stage3.st->p;
    const size_t nk = (size_t)rk + 1, parts2 = scan_partials_words((uint32_t)nk);
    mem3.init(device, S, need2 + (rv + 4 * rf + 2 * nk + parts2 + 64) * 4 + ((size_t)1 << 16));
    uint8_t* d2 = mem3.take<uint8_t>(need2);
    RelabelBatch b{};
    b.rank = mem3.take<uint32_t>(rv ? rv : 1); b.key = mem3.take<uint32_t>(rf ? rf : 1); b.count = mem3.take<uint32_t>(nk);
    b.order = mem3.take<uint32_t>(rf ? rf : 1); b.new_face = mem3.take<uint32_t>(rf ? rf : 1); b.scan_partials = mem3.take<uint32_t>(parts2);
    if (!d2 || !b.rank || !b.key || !b.count || !b.order || !b.new_face || !b.scan_partials) return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (batch relabelling)");
    std::vector<size_t> table_at(items.size(), 0);
    { size_t at = off_tables; for (size_t i = 0; i < items.size(); ++i) if (c2v_src[i]) { table_at[i] = at; at += (size_t)items[i].F * 24; } }
    if ((rc = parallel_over((uint32_t)items.size(), [&](uint32_t, uint32_t i) -> int {
          std::memcpy(h2 + (size_t)items[i].seq_off * 4, seq_src[i], (size_t)items[i].n_seq * 4);
          if (c2v_src[i]) { std::memcpy(h2 + table_at[i], c2v_src[i], (size_t)items[i].F * 12); std::memcpy(h2 + table_at[i] + (size_t)items[i].F * 12, opp_src[i], (size_t)items[i].F * 12); }
          return DMI_OK;
        }, [&](uint32_t i) { return (uint64_t)items[i].n_seq + (c2v_src[i] ? 6ull * items[i].F : 0ull); }))) return rc;
    for (size_t i = 0; i < items.size(); ++i) {
      RelabelItem& it = items[i];
      it.seq = reinterpret_cast<const uint32_t*>(d2) + it.seq_off;
      if (c2v_src[i]) { it.c2v = reinterpret_cast<const uint32_t*>(d2 + table_at[i]); it.opp = it.c2v + (size_t)it.F * 3; }
    }
    if (!items.empty()) std::memcpy(h2 + off_items, items.data(), items.size() * sizeof(RelabelItem));
    if (!fans.empty()) std::memcpy(h2 + off_fans, fans.data(), fans.size() * sizeof(FanItem));
    if (!comps.empty()) std::memcpy(h2 + off_comps, comps.data(), comps.size() * sizeof(ComposeItem));
    if (!copies.empty()) std::memcpy(h2 + off_copies, copies.data(), copies.size() * sizeof(CopyItem));
    if (!clears.empty()) std::memcpy(h2 + off_clears, clears.data(), clears.size() * sizeof(CopyItem));
    HIP_TRY(hipMemcpyAsync(d2, h2, need2, hipMemcpyHostToDevice, S));
    HIP_TRY(hipMemsetAsync(b.rank, 0xFF, (size_t)(rv ? rv : 1) * 4, S));
    HIP_TRY(hipMemsetAsync(b.count, 0, nk * 4, S));
    b.items = reinterpret_cast<const RelabelItem*>(d2 + off_items); b.n_items = (uint32_t)items.size();
    b.total_faces = (uint32_t)rf; b.total_verts = (uint32_t)rv; b.total_keys = (uint32_t)rk; b.total_seq = (uint32_t)rs; b.total_remap_faces = (uint32_t)rr;
    launch_clear_items(reinterpret_cast<const CopyItem*>(d2 + off_clears), (uint32_t)clears.size(), S);
    launch_scatter_items(reinterpret_cast<const CopyItem*>(d2 + off_copies), (uint32_t)copies.size(), base, S);
    launch_relabel_batch(b, S);
    launch_compose_batch(reinterpret_cast<const ComposeItem*>(d2 + off_comps), (uint32_t)comps.size(), (uint32_t)comp_total, S);
    launch_build_fans_batch(reinterpret_cast<const FanItem*>(d2 + off_fans), (uint32_t)fans.size(), (uint32_t)fan_total, S);
  }
  // (while the device works: the meshes' host tables go back to the pool — a thousand small frees — on the worker threads)
  (void)parallel_over(M, [&](uint32_t, uint32_t kk) -> int { owners[kk].reset(); return DMI_OK; }, nullptr);
  HIP_TRY(hipStreamSynchronize(S));
  const double t_dev = ms();
  groups.clear();
  if (trace) {
    uint64_t tf = 0;
    for (uint32_t k = 0; k < M; ++k) tf += meshes[which_all[k]].num_faces;
    std::fprintf(stderr, "[dmi] batch prepare, device form: %u meshes, %llu faces: layout + pack + issue of the connectivity kernels %.2f ms, host walks + job layouts %.2f (%u threads), "
                         "sequences up + relabelling + fan rows %.2f, release %.2f; %zu jobs deferred, total %.2f; thread time: waiting for tables %.1f ms, connectivity (Edgebreaker, sequencer, bytes) %.1f, job layout %.1f, output buffer %.1f\n", M, (unsigned long long)tf, t_issue, t_walks - t_issue, n_threads, t_dev - t_walks, ms() - t_dev, items.size(), ms(),
                         ns_wait.load() / 1e6, ns_conn.load() / 1e6, ns_job.load() / 1e6, ns_buf.load() / 1e6);
    std::fprintf(stderr, "[dmi]   connectivity thread time: attribute tables %.1f ms, Edgebreaker + connectivity bytes %.1f, universal sequencer (+ views) %.1f, rest %.1f\n",
                 g_conn_us[0].exchange(0) / 1e3, g_conn_us[1].exchange(0) / 1e3, g_conn_us[2].exchange(0) / 1e3, g_conn_us[3].exchange(0) / 1e3);
    std::fprintf(stderr, "[dmi]   Edgebreaker thread time by step (%llu meshes): set-up %.2f ms, traversal %.2f, symbols → bits %.2f, seam streams %.2f; whole call incl. its destructors %.2f\n", (unsigned long long)g_eb_ns[4].exchange(0),
                 g_eb_ns[0].exchange(0) / 1e6, g_eb_ns[1].exchange(0) / 1e6, g_eb_ns[2].exchange(0) / 1e6, g_eb_ns[3].exchange(0) / 1e6, g_eb_ns[5].exchange(0) / 1e6);
  }
  return DMI_OK;
}

// worker t's stream on `device` (process-lifetime pool, created on first use)
static std::shared_ptr<StreamHolder> prepare_worker_stream(uint32_t t, int device) {
  static std::mutex m;
  static std::vector<std::pair<int, std::shared_ptr<StreamHolder>>> pool[kMaxPrepareWorkers];
  std::lock_guard<std::mutex> lock(m);
  std::shared_ptr<StreamHolder> found;
  for (auto& e : pool[t]) if (e.first == device) found = e.second;
  if (!found && hipSetDevice(device) == hipSuccess) {
    found = std::make_shared<StreamHolder>();
    if (hipStreamCreate(&found->s) != hipSuccess) found.reset(); else pool[t].push_back({device, found});
  }
  return found;
}

static int meshes_prepare_impl(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, const int32_t* device_of_mesh, dmi_buffer* header_and_connectivity, dmi_job** jobs) {
  if (!meshes || !header_and_connectivity || !jobs || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  for (uint32_t j = 0; j < n; ++j) { jobs[j] = nullptr; header_and_connectivity[j] = dmi_buffer{}; }
  // (the walks are serial per mesh and independent across meshes: as many workers as the host gives — 128 at most — minus what a
  //  concurrent dmi_jobs_encode of the previous batch needs; DMI_HOST_THREADS caps a process's share)
  const uint32_t n_threads = std::max(1u, std::min({n, (uint32_t)host_threads(), kMaxPrepareWorkers}));
  std::vector<int> rcs(n, DMI_OK);
  std::vector<std::string> errs(n);
  std::atomic<uint32_t> next{0};
  // largest meshes first: the walks are serial per mesh, so the longest one should not start last
  std::vector<uint32_t> order(n);
  for (uint32_t j = 0; j < n; ++j) order[j] = j;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return meshes[x].num_faces > meshes[y].num_faces; });
  const bool library_streams = !(cfg && cfg->stream);
  const std::function<std::shared_ptr<StreamHolder>(uint32_t, int)> worker_stream = prepare_worker_stream;
  // device form first (prepare_slice_device): the eligible meshes of every device in slices of ≤ 64M faces; whatever it leaves goes mesh by mesh below
  std::vector<uint8_t> done(n, 0);
  int ndev = 0;
  if (!std::getenv("DMI_HOST_CONNECTIVITY") && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
    std::vector<int> devices;
    for (uint32_t j = 0; j < n; ++j) { const int d = device_of_mesh ? device_of_mesh[j] : (cfg ? cfg->device : 0); if (std::find(devices.begin(), devices.end(), d) == devices.end()) devices.push_back(d); }
    const uint32_t min_faces = std::getenv("DMI_BATCH_MIN_FACES") ? (uint32_t)std::atoi(std::getenv("DMI_BATCH_MIN_FACES")) : 1u;
    auto eligible = [&](const dmi_mesh& m) {
      if (!m.atts || m.num_atts == 0 || m.num_atts > 255 || !m.faces || m.num_faces < min_faces || m.num_faces >= kDeviceRelabelMinFaces) return false;
      if (m.atts[0].att_type != DMI_ATT_POSITION || m.atts[0].num_unique == 0 || m.atts[0].num_points == 0) return false;
      for (uint32_t i = 0; i < m.num_atts; ++i) if (m.atts[i].num_unique && !m.atts[i].values) return false;
      return true;
    };
    std::vector<int> dev_rc(devices.size(), DMI_OK);
    std::vector<std::string> dev_err(devices.size());
    auto run_device = [&](size_t g) {
      const int d = devices[g];
      dmi_config c{};
      if (cfg) c = *cfg;
      c.device = d;
      const uint32_t share = std::max(1u, n_threads / (uint32_t)devices.size());
      std::vector<uint32_t> slice;
      uint64_t faces = 0, verts = 0;
      auto flush = [&]() -> int {
        if (slice.empty()) return DMI_OK;
        const int rc = prepare_slice_device(meshes, slice, c, d, share, header_and_connectivity, jobs, done, worker_stream);
        slice.clear(); faces = verts = 0;
        return rc;
      };
      for (uint32_t j = 0; j < n && !dev_rc[g]; ++j) {
        if ((device_of_mesh ? device_of_mesh[j] : c.device) != d || !eligible(meshes[j])) continue;
        if (faces + meshes[j].num_faces > (64u << 20) || verts + meshes[j].atts[0].num_unique >= (1u << 30)) { if ((dev_rc[g] = flush())) break; }
        slice.push_back(j); faces += meshes[j].num_faces; verts += meshes[j].atts[0].num_unique;
      }
      if (!dev_rc[g]) dev_rc[g] = flush();
      if (dev_rc[g]) dev_err[g] = g_last_error;
    };
    if (devices.size() == 1) run_device(0);
    else { std::vector<std::thread> th; for (size_t g = 0; g < devices.size(); ++g) th.emplace_back(run_device, g); for (auto& x : th) x.join(); }
    for (size_t g = 0; g < devices.size(); ++g)
      if (dev_rc[g]) {
        for (uint32_t k = 0; k < n; ++k) { if (jobs[k]) { dmi_job_destroy(jobs[k]); jobs[k] = nullptr; } dmi_free(&header_and_connectivity[k]); }
        return fail(dev_rc[g], dev_err[g]);
      }
  }
  auto work = [&](uint32_t t) {
    int adopted_for = -1;
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= n) break;
      const uint32_t j = order[k];
      if (done[j]) continue;
      dmi_config c{};
      if (cfg) c = *cfg;
      if (device_of_mesh) c.device = device_of_mesh[j];
      // (creating a stream costs ≈ 1 ms and serialises across threads: the workers share kPrepareStreams of them)
      if (library_streams && adopted_for != c.device) { g_adopt_stream = worker_stream(t % kPrepareStreams, c.device); adopted_for = c.device; }   // (null: dmi_job_create makes its own)
      if (jobs[j]) { dmi_job_destroy(jobs[j]); jobs[j] = nullptr; }
      dmi_free(&header_and_connectivity[j]);
      rcs[j] = dmi_mesh_prepare(&meshes[j], &c, &header_and_connectivity[j], &jobs[j]);
      if (rcs[j]) errs[j] = g_last_error;
    }
    g_adopt_stream.reset();
  };
  bool any_left = false;
  for (uint32_t j = 0; j < n; ++j) any_left = any_left || !done[j];
  if (!any_left) {}
  else if (n_threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (uint32_t j = 0; j < n; ++j) {
    if (!rcs[j]) continue;
    const int rc = rcs[j];
    const std::string e = "mesh " + std::to_string(j) + ": " + errs[j];
    for (uint32_t k = 0; k < n; ++k) { if (jobs[k]) { dmi_job_destroy(jobs[k]); jobs[k] = nullptr; } dmi_free(&header_and_connectivity[k]); }
    return fail(rc, e);
  }
  return DMI_OK;
}

// The host-core stream coders of the hybrid form on their own (no device involved): tests pin them against the oracle's coders,
// bench.py times them on one core of the GPU box beside the device walker.
int dmi_host_rans_stream(const uint32_t* freq, uint32_t num_symbols, uint32_t precision, const uint32_t* symbols, uint64_t n, dmi_buffer* out) {
  if (!freq || !out || (!symbols && n) || precision < 8 || precision > 20) return fail(DMI_ERR_INVALID_ARGUMENT, "bad argument");
  std::vector<RansEntry> table(num_symbols);
  uint64_t cum = 0;
  for (uint32_t k = 0; k < num_symbols; ++k) {
    if (freq[k] > (1u << precision)) return fail(DMI_ERR_INVALID_ARGUMENT, "frequency above 2^precision");
    table[k] = make_rans_entry(freq[k], (uint32_t)cum, precision);
    cum += freq[k];
  }
  if (cum != (1ull << precision)) return fail(DMI_ERR_INVALID_ARGUMENT, "frequencies must sum to 2^precision");
  for (uint64_t k = 0; k < n; ++k) if (symbols[k] >= num_symbols || !freq[symbols[k]]) return fail(DMI_ERR_ENTROPY, "symbol without a frequency");
  HostChainOut o;
  host_rans_chain(symbols, n, table.data(), num_symbols, precision, o);
  if (o.err) return fail(o.err == 2 ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_ENTROPY, o.err == 1 ? "rANS state too large" : "host chain error");
  out->data = o.data; out->len = o.len; out->cap = o.cap;
  o.data = nullptr; o.cap = 0;   // ownership moves to the caller (dmi_free → free)
  return DMI_OK;
}
int dmi_host_rabs_stream(uint8_t zero_prob, const uint8_t* bits, uint64_t n, dmi_buffer* out) {
  if (!out || (!bits && n) || zero_prob == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "bad argument");
  const uint32_t p0 = zero_prob, f1 = 256u - p0;
  const RansEntry e[2] = {make_rans_entry(p0, f1, 8), make_rans_entry(f1, 0, 8)};
  HostChainOut o;
  host_rabs_chain(bits, n, e, o);
  if (o.err) return fail(o.err == 2 ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_ENTROPY, o.err == 1 ? "rABS state too large" : "host chain error");
  out->data = o.data; out->len = o.len; out->cap = o.cap;
  o.data = nullptr; o.cap = 0;
  return DMI_OK;
}

int dmi_host_rabs_constant_stream(uint8_t zero_prob, uint32_t bit, uint64_t n, dmi_buffer* out) {
  if (!out || zero_prob == 0 || bit > 1) return fail(DMI_ERR_INVALID_ARGUMENT, "bad argument");
  std::vector<uint8_t> bytes;
  if (!host_rabs_constant(zero_prob, bit, n, bytes)) return fail(DMI_ERR_ENTROPY, "rABS state too large");
  return to_buffer(bytes, out);
}

int dmi_encode_mesh(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* out) {
  if (!out) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  NumaPin pin(cfg ? cfg->device : 0);   // (also around the encode: the host-core chains read the symbols the staging DMA brought in)
  dmi_buffer head{}, att{};
  dmi_job* job = nullptr;
  const bool trace = std::getenv("DMI_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  struct OneShot { OneShot() { g_one_shot_call = true; } ~OneShot() { g_one_shot_call = false; } } one_shot;
  int rc = dmi_mesh_prepare(mesh, cfg, &head, &job);
  if (rc) return rc;
  const double t_prep = ms();
  rc = dmi_job_encode(job, &att);
  const double t_enc = ms();
  { const dmi_timings pre = g_last_call; g_last_call = job->last; g_last_call.tables_ms = pre.tables_ms; g_last_call.connectivity_ms = pre.connectivity_ms; g_last_call.job_create_ms = pre.job_create_ms; g_last_call.job_create_device_ms = pre.job_create_device_ms; }
  dmi_job_destroy(job);
  const double t_destroy = ms();
  if (rc) { dmi_free(&head); return rc; }
  // header + connectivity + attribute section in one library-owned buffer
  out->data = static_cast<uint8_t*>(std::malloc(head.len + att.len ? head.len + att.len : 1));
  if (!out->data) { dmi_free(&head); dmi_free(&att); return fail(DMI_ERR_OUT_OF_MEMORY, "out of host memory"); }
  std::memcpy(out->data, head.data, head.len);
  std::memcpy(out->data + head.len, att.data, att.len);
  out->len = out->cap = head.len + att.len;
  dmi_free(&head);
  dmi_free(&att);
  g_last_call.call_ms = (float)ms();
  if (trace) std::fprintf(stderr, "[dmi] encode_mesh: prepare %.1f ms, encode %.1f, job destroy %.1f, splice %.1f\n", t_prep, t_enc - t_prep, t_destroy - t_enc, ms() - t_destroy);
  return DMI_OK;
}

// encode::encode for a mesh whose buffers already live in HBM: `mesh` is a host struct whose faces, attribute values and point → value maps
// are DEVICE pointers on cfg->device.  Faces and maps are read back once (the Edgebreaker traversal and the sequencer are serial host walks),
// the values never leave the device; the tables are built from the device faces where they are.
int dmi_encode_mesh_device(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* out) {
  if (!out || !mesh || !mesh->atts || mesh->num_atts == 0 || mesh->num_atts > 255 || (!mesh->faces && mesh->num_faces)) return fail(DMI_ERR_INVALID_ARGUMENT, "bad mesh");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  const int device = cfg ? cfg->device : 0;
  HIP_TRY(hipSetDevice(device));
  NumaPin pin(device);
  auto holder = thread_stream(device);
  if (!holder) return fail(DMI_ERR_HIP, "hipStreamCreate");
  hipStream_t s = cfg && cfg->stream ? static_cast<hipStream_t>(cfg->stream) : holder->s;
  const bool trace = std::getenv("DMI_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  // host shadow: faces and the distinct maps in one pinned read-back (huge pages: the walks read the faces from there)
  const size_t C = (size_t)mesh->num_faces * 3;
  std::vector<dmi_attribute> atts(mesh->atts, mesh->atts + mesh->num_atts);
  std::vector<size_t> map_at(mesh->num_atts, (size_t)-1);
  size_t need = ((C * 4 + 255) & ~(size_t)255);
  for (uint32_t i = 0; i < mesh->num_atts; ++i) {
    if (!atts[i].point_to_value) continue;
    for (uint32_t j = 0; j < i; ++j) if (mesh->atts[j].point_to_value == atts[i].point_to_value) map_at[i] = map_at[j];
    if (map_at[i] == (size_t)-1) { map_at[i] = need; need += ((size_t)atts[i].num_points * 4 + 255) & ~(size_t)255; }
  }
  struct StageGuard { HostStage* st = nullptr; ~StageGuard() { release_stage(st); } } stage;
  stage.st = acquire_stage(device, need + 256);
  if (!stage.st) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (mesh read-back)");
  uint8_t* hp = stage.st->p;
  // (on a stream of its own: the table kernels of mesh_prepare_impl run on the thread's stream meanwhile; the host waits for both before its walks)
  hipStream_t s_down = cfg && cfg->stream ? s : group_stream(device, 0);
  if (!s_down) s_down = s;
  if (C) HIP_TRY(hipMemcpyAsync(hp, mesh->faces, C * 4, hipMemcpyDeviceToHost, s_down));
  for (uint32_t i = 0; i < mesh->num_atts; ++i) {
    if (!atts[i].point_to_value) continue;
    bool first = true;
    for (uint32_t j = 0; j < i; ++j) if (mesh->atts[j].point_to_value == mesh->atts[i].point_to_value) first = false;
    if (first) HIP_TRY(hipMemcpyAsync(hp + map_at[i], mesh->atts[i].point_to_value, (size_t)atts[i].num_points * 4, hipMemcpyDeviceToHost, s_down));
    atts[i].point_to_value = reinterpret_cast<const uint32_t*>(hp + map_at[i]);
  }
  struct DownEvent { hipEvent_t e = nullptr; ~DownEvent() { g_faces_event = nullptr; if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); } } } down;
  HIP_TRY(hipEventCreateWithFlags(&down.e, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(down.e, s_down));
  g_faces_event = down.e;
  const double t_down = ms();   // (issue time: the copy itself runs beside the table kernels and is waited for inside mesh_prepare_impl)
  dmi_mesh shadow{reinterpret_cast<const uint32_t*>(hp), mesh->num_faces, atts.data(), mesh->num_atts};
  const DeviceMeshSrc src{mesh->faces, mesh->atts[0].point_to_value};
  dmi_buffer head{}, att{};
  dmi_job* job = nullptr;
  struct OneShot { OneShot() { g_one_shot_call = true; } ~OneShot() { g_one_shot_call = false; } } one_shot;
  int rc = mesh_prepare_impl(&shadow, cfg, &head, &job, &src);
  std::vector<std::vector<uint8_t>> host_values;
  if (rc == kNeedHostValues) {   // outside the order-free class: the reference's serial walks and the host relabelling read everything on the host
    host_values.resize(mesh->num_atts);
    for (uint32_t i = 0; i < mesh->num_atts; ++i) {
      const size_t vb = (size_t)atts[i].num_unique * atts[i].num_components * 4;
      host_values[i].resize(vb ? vb : 1);
      if (vb) HIP_TRY(hipMemcpy(host_values[i].data(), atts[i].values, vb, hipMemcpyDeviceToHost));
      atts[i].values = host_values[i].data();
    }
    rc = mesh_prepare_impl(&shadow, cfg, &head, &job, nullptr);
  }
  if (rc) return rc;
  const double t_prep = ms();
  rc = dmi_job_encode(job, &att);
  { const dmi_timings pre = g_last_call; g_last_call = job->last; g_last_call.tables_ms = pre.tables_ms; g_last_call.connectivity_ms = pre.connectivity_ms; g_last_call.job_create_ms = pre.job_create_ms; g_last_call.job_create_device_ms = pre.job_create_device_ms; g_last_call.mesh_readback_ms = (float)t_down; }
  dmi_job_destroy(job);
  if (rc) { dmi_free(&head); return rc; }
  out->data = static_cast<uint8_t*>(std::malloc(head.len + att.len ? head.len + att.len : 1));
  if (!out->data) { dmi_free(&head); dmi_free(&att); return fail(DMI_ERR_OUT_OF_MEMORY, "out of host memory"); }
  std::memcpy(out->data, head.data, head.len);
  std::memcpy(out->data + head.len, att.data, att.len);
  out->len = out->cap = head.len + att.len;
  dmi_free(&head);
  dmi_free(&att);
  g_last_call.call_ms = (float)ms();
  if (trace) std::fprintf(stderr, "[dmi] encode_mesh_device: faces + maps read back %.1f ms, prepare %.1f, encode + splice %.1f\n", t_down, t_prep - t_down, ms() - t_prep);
  return DMI_OK;
}

// ---- dmi_meshes_prepare for device-built meshes (dmi_meshes_build): nothing is packed or uploaded again ----
// One member of a built group through the single-mesh path: its tables from the device copies of its faces / position map, its values copied
// device to device; a mesh the order-free table construction does not cover takes the host walks with its values brought down first.
static int prepare_built_single(const dmi_mesh& view, const BuiltDevice& bd, const dmi_config& cfg, dmi_buffer* head, dmi_job** job) {
  const BuiltGroup& bg = *bd.group;
  const BuiltGroup::Member& mem = bg.members[bd.member];
  std::vector<dmi_attribute> atts(view.atts, view.atts + view.num_atts);
  for (uint32_t i = 0; i < view.num_atts; ++i) atts[i].values = mem.atts[i].n_unique ? static_cast<const void*>(bg.d_base + mem.atts[i].val_off) : nullptr;
  const dmi_mesh shadow{view.faces, view.num_faces, atts.data(), view.num_atts};
  const DeviceMeshSrc src{reinterpret_cast<const uint32_t*>(bg.d_base + mem.faces_off),
                          mem.atts[0].map_off != (size_t)-1 ? reinterpret_cast<const uint32_t*>(bg.d_base + mem.atts[0].map_off) : nullptr};
  dmi_config c = cfg;
  c.device = bg.device;
  int rc = mesh_prepare_impl(&shadow, &c, head, job, &src);
  if (rc != kNeedHostValues) return rc;
  std::vector<std::vector<uint8_t>> host_values(view.num_atts);
  HIP_TRY(hipSetDevice(bg.device));
  for (uint32_t i = 0; i < view.num_atts; ++i) {
    if (view.atts[i].values) { atts[i].values = view.atts[i].values; continue; }   // (read back by the build: DMI_BUILD_HOST_VALUES)
    const size_t vb = (size_t)atts[i].num_unique * atts[i].num_components * 4;
    host_values[i].resize(vb ? vb : 1);
    if (vb) HIP_TRY(hipMemcpy(host_values[i].data(), bg.d_base + mem.atts[i].val_off, vb, hipMemcpyDeviceToHost));
    atts[i].values = host_values[i].data();
  }
  return mesh_prepare_impl(&shadow, &c, head, job, nullptr);
}

int dmi_built_meshes_prepare(const dmi_built_mesh* built, uint32_t n, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** jobs) {
  if (!built || !header_and_connectivity || !jobs || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  for (uint32_t j = 0; j < n; ++j) { jobs[j] = nullptr; header_and_connectivity[j] = dmi_buffer{}; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  dmi_config c0{};
  if (cfg) c0 = *cfg;
  if (c0.stream) return fail(DMI_ERR_INVALID_ARGUMENT, "dmi_built_meshes_prepare runs on library streams: leave dmi_config.stream null");
  const uint32_t n_threads = std::max(1u, std::min({n, (uint32_t)host_threads(), kMaxPrepareWorkers}));
  std::vector<dmi_mesh> views(n);
  std::vector<const BuiltDevice*> dev(n, nullptr);
  std::vector<uint32_t> host_list;
  for (uint32_t j = 0; j < n; ++j) {
    views[j] = built[j].mesh;
    if (!built[j].owner) return fail(DMI_ERR_INVALID_ARGUMENT, "built mesh " + std::to_string(j) + " is empty");
    dev[j] = dynamic_cast<const BuiltDevice*>(static_cast<const BuiltBase*>(built[j].owner));
    if (!dev[j]) host_list.push_back(j);
  }
  auto bail = [&](int rc) {
    const std::string e = g_last_error;
    for (uint32_t k = 0; k < n; ++k) { if (jobs[k]) { dmi_job_destroy(jobs[k]); jobs[k] = nullptr; } dmi_free(&header_and_connectivity[k]); }
    return fail(rc, e);
  };
  int rc;
  // meshes the host builder made (outside the device form's class): the ordinary batch prepare
  if (!host_list.empty()) {
    std::vector<dmi_mesh> hm;
    for (uint32_t j : host_list) hm.push_back(views[j]);
    std::vector<dmi_buffer> hh(hm.size());
    std::vector<dmi_job*> hj(hm.size(), nullptr);
    if ((rc = meshes_prepare_impl(hm.data(), (uint32_t)hm.size(), &c0, nullptr, hh.data(), hj.data()))) return bail(rc);
    for (size_t k = 0; k < host_list.size(); ++k) { header_and_connectivity[host_list[k]] = hh[k]; jobs[host_list[k]] = hj[k]; }
  }
  // the device-built ones, group by group in the order of their first member
  std::vector<uint8_t> done(n, 0);
  std::vector<BuiltGroup*> order;
  for (uint32_t j = 0; j < n; ++j) if (dev[j] && std::find(order.begin(), order.end(), dev[j]->group.get()) == order.end()) order.push_back(dev[j]->group.get());
  std::vector<uint32_t> singles;
  std::vector<AdoptedGroup> slice;
  std::vector<uint32_t> slice_which;
  uint64_t slice_faces = 0, slice_verts = 0;
  int slice_device = -1;
  auto flush = [&]() -> int {
    if (slice.empty()) return DMI_OK;
    dmi_config c = c0;
    c.device = slice_device;
    const int r = prepare_slice_device(views.data(), slice_which, c, slice_device, n_threads, header_and_connectivity, jobs, done, prepare_worker_stream, &slice);
    slice.clear(); slice_which.clear(); slice_faces = slice_verts = 0;
    return r;
  };
  for (BuiltGroup* bg : order) {
    AdoptedGroup ag{bg, std::vector<int32_t>(bg->members.size(), -1)};
    for (uint32_t j = 0; j < n; ++j) if (dev[j] && dev[j]->group.get() == bg) {
      if (dev[j]->member >= ag.present.size() || ag.present[dev[j]->member] >= 0) return bail(fail(DMI_ERR_INVALID_ARGUMENT, "built mesh " + std::to_string(j) + " appears twice"));
      ag.present[dev[j]->member] = (int32_t)j;
    }
    bool large = false;
    uint64_t verts = 0;
    for (const auto& mem : bg->members) { large = large || mem.F >= kDeviceRelabelMinFaces; verts += mem.atts.empty() ? 0u : mem.atts[0].n_unique; }
    if (large || std::getenv("DMI_HOST_CONNECTIVITY")) { for (int32_t j : ag.present) if (j >= 0) singles.push_back((uint32_t)j); continue; }
    if (!slice.empty() && (slice_device != bg->device || slice_faces + bg->total_faces > (64u << 20) || slice_verts + verts >= (1u << 30)) && (rc = flush())) return bail(rc);
    slice_device = bg->device;
    for (int32_t j : ag.present) if (j >= 0) slice_which.push_back((uint32_t)j);
    slice_faces += bg->total_faces; slice_verts += verts;
    slice.push_back(std::move(ag));
  }
  if ((rc = flush())) return bail(rc);
  // what the batched form left: meshes its table kernels flagged (the reference's serial walks decide) and the large ones
  for (uint32_t j = 0; j < n; ++j) if (dev[j] && !done[j] && std::find(singles.begin(), singles.end(), j) == singles.end()) singles.push_back(j);
  if (!singles.empty()) {
    std::atomic<uint32_t> next{0};
    const uint32_t nt = std::max(1u, std::min(n_threads, (uint32_t)singles.size()));
    std::vector<int> rcs(singles.size(), DMI_OK);
    std::vector<std::string> errs(singles.size());
    auto work = [&](uint32_t t) {
      for (uint32_t k; (k = next.fetch_add(1)) < singles.size();) {
        const uint32_t j = singles[k];
        if (jobs[j]) { dmi_job_destroy(jobs[j]); jobs[j] = nullptr; }
        dmi_free(&header_and_connectivity[j]);
        g_adopt_stream = prepare_worker_stream(t % kPrepareStreams, dev[j]->group->device);
        rcs[k] = prepare_built_single(views[j], *dev[j], c0, &header_and_connectivity[j], &jobs[j]);
        g_adopt_stream.reset();
        if (rcs[k]) errs[k] = "mesh " + std::to_string(j) + ": " + g_last_error;
      }
    };
    if (nt == 1) work(0);
    else { std::vector<std::thread> th; for (uint32_t t = 0; t < nt; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
    for (size_t k = 0; k < singles.size(); ++k) if (rcs[k]) { g_last_error = errs[k]; return bail(rcs[k]); }
  }
  return DMI_OK;
}

}  // extern "C"
