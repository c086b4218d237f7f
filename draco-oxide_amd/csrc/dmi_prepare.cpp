// dmi_prepare.cpp — whole-mesh entry points of the C ABI: the connectivity stage (host walks + device tables), dmi_mesh_prepare /
// dmi_meshes_prepare / dmi_encode_mesh, and the host-core stream coders on their own.
#include "dmi_job.hpp"

using namespace dmi;

namespace {

// A library stream per (host thread, device) for the connectivity stage of a whole-mesh call — and, adopted, for the job it creates
// (hipStreamCreate costs ≈ 1 ms and serialises across threads).
std::shared_ptr<StreamHolder> thread_stream(int device) {
  static thread_local std::vector<std::pair<int, std::shared_ptr<StreamHolder>>> mine;
  for (auto& e : mine) if (e.first == device) return e.second;
  auto h = std::make_shared<StreamHolder>();
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  mine.push_back({device, h});
  return h;
}

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
  const uint32_t *h_opp = nullptr, *h_c2v = nullptr;
  uint32_t* h_lmc = nullptr;
  const uint8_t* h_onb = nullptr;
  uint32_t V = 0, Vcap = 0, flags = 0;
  bool valid = false, have_lmc = false;
  double t_up = 0, t_kernels = 0, t_down = 0;
  ~DeviceTables() { if (stream && valid) (void)hipStreamSynchronize(stream); release_stage(host); }

  // DMI_OK with valid = true: ct views the tables; DMI_OK with valid = false: not covered (the caller runs the host builder); else an error
  int build(const dmi_mesh* mesh, CornerTables& ct) {
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
    d_faces = mem.take<uint32_t>(C);
    uint32_t* d_p2v = mapped ? mem.take<uint32_t>(P) : nullptr;
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
    h_lmc = reinterpret_cast<uint32_t*>(hp + C * 4 * (mapped ? 2 : 1));
    uint8_t* hp_onb = reinterpret_cast<uint8_t*>(h_lmc + nv);
    uint32_t* hp_words = reinterpret_cast<uint32_t*>(hp + ((C * 4 * (mapped ? 2 : 1) + nv * 5 + 255) & ~(size_t)255));
    const ConnMeshDesc desc{0u, 0u, F, Vcap, mapped ? 0u : kNone, P, 0u, 0u};
    HIP_TRY(hipMemcpyAsync(d_desc, &desc, sizeof desc, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_faces, mesh->faces, C * 4, hipMemcpyHostToDevice, stream));
    if (mapped) HIP_TRY(hipMemcpyAsync(d_p2v, pos.point_to_value, (size_t)P * 4, hipMemcpyHostToDevice, stream));
    t_up = ms();
    a.meshes = d_desc; a.M = 1; a.total_faces = F; a.total_verts = Vcap;
    a.faces = d_faces; a.p2v = d_p2v; a.c2v = d_c2v; a.opp = d_opp; a.lmc = d_lmc; a.on_boundary = d_onb; a.flags = d_words; a.vmax = d_words + 1;
    HIP_TRY(conn_tables_clear(a, stream));
    launch_conn_tables(a, stream);
    HIP_TRY(hipMemcpyAsync(hp_words, d_words, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(hp_opp, d_opp, C * 4, hipMemcpyDeviceToHost, stream));
    if (mapped) HIP_TRY(hipMemcpyAsync(hp_c2v, d_c2v, C * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(hp_onb, d_onb, Vcap, hipMemcpyDeviceToHost, stream));
    t_kernels = ms();
    // while the device works: the vertex ids of a mesh without a position map are its faces — the walks read them at random, so they get a
    // copy on huge pages (the caller's array is on whatever pages its allocator chose); storage from the host pool, written in parallel slices
    const uint32_t* c2v_host = nullptr;
    if (!mapped) {
      pool_fit(ct.c2v_own, C);
      if (ct.c2v_own.capacity() < C) ct.c2v_own.reserve(C);   // (below the pool's size threshold pool_fit hands out an empty vector)
      uint32_t* dst = ct.c2v_own.data();   // (capacity ≥ C; the vector's size stays 0: it only carries the storage back to the pool)
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
    h_opp = hp_opp; h_c2v = hp_c2v; h_onb = hp_onb;
    ct.F = F; ct.V = V;
    ct.c2p = mesh->faces;
    ct.c2v = mapped ? h_c2v : c2v_host;
    ct.opp = h_opp;
    ct.lmc = nullptr;   // fetch_lmc() when a reader needs it
    ct.no_boundary = !(flags & CONN_HAS_BOUNDARY);
    ct.att.clear();
    valid = true;
    return DMI_OK;
  }
  int fetch_lmc(CornerTables& ct) {
    if (!valid || have_lmc) return DMI_OK;
    HIP_TRY(hipMemcpyAsync(h_lmc, d_lmc, (size_t)V * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    ct.lmc = h_lmc;
    have_lmc = true;
    return DMI_OK;
  }
};

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
    pool_give(eb.seeds);
    for (auto& q : seqs) pool_give(q);
  }
  std::vector<dmi_corner_table> views;
};

static int build_connectivity(const dmi_mesh* mesh, ConnOwner& o, std::vector<uint8_t>& bytes, DeviceTables* dt = nullptr) {
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
  if (dt && (rc = dt->build(mesh, o.ct))) return rc;   // (the device pass range-checks the faces and the position map itself)
  const bool on_device = dt && dt->valid;
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
    rc = o.ct.build_universal(mesh->faces, mesh->num_faces, mesh->atts[0].point_to_value, err, /*copy_faces=*/!dt);
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
  const bool overlap = mesh->num_faces > 100000;
  double t_att = 0, t_eb = 0, t_seq = 0;
  o.views.resize(mesh->num_atts);
  o.seqs.resize(mesh->num_atts);
  auto universal_view = [&](dmi_corner_table& v) { v.num_vertices = o.ct.V; v.corner_to_vertex = o.ct.c2v; v.opposite = o.ct.opp; v.left_most_corner = o.ct.lmc; };
  Pooled<uint8_t> on_boundary_p;
  std::vector<uint8_t>& on_boundary = on_boundary_p.v;
  const uint8_t* boundary_flags = on_device ? dt->h_onb : nullptr;   // per vertex: on a boundary of the universal table (the device pass computes them with the left-most corners)
  auto sequence_universal = [&] {
    TableRef tr{o.ct.F, o.ct.V, o.ct.c2v, o.ct.opp, o.ct.lmc};
    attribute_sequence(tr, o.eb.seeds.data(), (uint32_t)o.eb.seeds.size(), o.seqs[0], boundary_flags ? boundary_flags : (on_boundary.empty() ? nullptr : on_boundary.data()));
  };
  if (on_device) {   // the host attribute-table builder walks fans from the left-most corners: fetched only for meshes with an attribute indexed unlike the Position attribute
    bool need_lmc = false;
    for (const uint32_t* m : maps) need_lmc = need_lmc || m != mesh->atts[0].point_to_value;
    if (need_lmc && (rc = dt->fetch_lmc(o.ct))) return rc;
  }
  auto build_att_tables = [&] {
    const auto a0 = tick();
    // an attribute indexed like the Position attribute has no seams but the boundary; one indexed like an earlier attribute has that one's table
    auto build_one = [&](size_t k) {
      for (size_t j = 0; j < k; ++j) if (maps[j] == maps[k]) return;   // (copied below, once its original is complete)
      o.ct.build_attribute_into(o.ct.att[k], maps[k], maps[k] == mesh->atts[0].point_to_value);
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
        auto walk = [&o, i, &t] { TableRef tr{o.ct.F, t.num_vertices, t.c2v.data(), t.opp.data(), t.lmc.data()}; attribute_sequence(tr, o.eb.seeds.data(), (uint32_t)o.eb.seeds.size(), o.seqs[i]); };
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
  if (trace && on_device) std::fprintf(stderr, "[dmi]   universal table of %u faces on the device: uploads issued %.2f ms, kernels + read-back issued %.2f, arrived %.2f (flags %#x)\n", mesh->num_faces, dt->t_up, dt->t_kernels, dt->t_down, dt->flags);
  if (trace) std::fprintf(stderr, "[dmi] host connectivity of %u faces (%s): universal corner table %.1f ms, attribute tables %.1f, Edgebreaker %.1f, universal sequencer %.1f, seam-table sequencers + views %.1f; total %.1f\n",
                          mesh->num_faces, overlap ? "overlapped: attribute tables and sequencer beside the Edgebreaker walk" : "in sequence", t_univ, t_att, t_eb, t_seq, since(c2), since(c0));
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
  CornerTables ct;
  *flags = 0; *num_vertices = 0;
  const int rc = dt.build(mesh, ct);
  *flags = dt.flags;
  if (rc) return rc;
  if (!dt.valid) return DMI_OK;   // flags say why: the host builder's case
  int rc2 = dt.fetch_lmc(ct);
  if (rc2) return rc2;
  *num_vertices = dt.V;
  std::memcpy(opposite, dt.h_opp, (size_t)mesh->num_faces * 12);
  if (left_most_corner) std::memcpy(left_most_corner, dt.h_lmc, (size_t)dt.V * 4);
  if (on_boundary) std::memcpy(on_boundary, dt.h_onb, dt.V);
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

int dmi_mesh_prepare(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** job) {
  if (!header_and_connectivity || !job) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
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
    const bool want_device = mesh && !std::getenv("DMI_HOST_CONNECTIVITY") && mesh->num_faces >= (in_batch ? kDeviceRelabelMinFaces : kDeviceTablesMinFaces) &&
                             hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    if (want_device) {
      dt.device = cfg ? cfg->device : 0;
      if (cfg && cfg->stream) dt.stream = static_cast<hipStream_t>(cfg->stream);
      else {
        if (!g_adopt_stream) { g_adopt_stream = thread_stream(dt.device); adopt.set = (bool)g_adopt_stream; }   // the job created below shares the stream
        if (g_adopt_stream) dt.stream = g_adopt_stream->s;
      }
    }
    rc = build_connectivity(mesh, o, bytes, want_device && dt.stream ? &dt : nullptr);
    if (rc) return rc;
    t_conn = ms();
    const DeviceTableView view{dt.d_faces, dt.d_c2v, dt.d_opp, true};
    rc = job_create_impl(mesh->atts, o.views.data(), mesh->num_atts, o.eb.seeds.data(), (uint32_t)o.eb.seeds.size(), cfg, dt.valid ? &view : nullptr, job);
    if (rc) return rc;
    t_create = ms();
    rc = to_buffer(bytes, header_and_connectivity);
    if (rc) { dmi_job_destroy(*job); *job = nullptr; }
    t_buf = ms();
  }
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
  auto worker_stream = [&](uint32_t t, int device) {   // worker t's stream on `device` (process-lifetime pool, created on first use)
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
  };
  auto work = [&](uint32_t t) {
    int adopted_for = -1;
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= n) break;
      const uint32_t j = order[k];
      dmi_config c{};
      if (cfg) c = *cfg;
      if (device_of_mesh) c.device = device_of_mesh[j];
      // (creating a stream costs ≈ 1 ms and serialises across threads: the workers share kPrepareStreams of them)
      if (library_streams && adopted_for != c.device) { g_adopt_stream = worker_stream(t % kPrepareStreams, c.device); adopted_for = c.device; }   // (null: dmi_job_create makes its own)
      rcs[j] = dmi_mesh_prepare(&meshes[j], &c, &header_and_connectivity[j], &jobs[j]);
      if (rcs[j]) errs[j] = g_last_error;
    }
    g_adopt_stream.reset();
  };
  if (n_threads == 1) work(0);
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

int dmi_encode_mesh(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* out) {
  if (!out) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  dmi_buffer head{}, att{};
  dmi_job* job = nullptr;
  const bool trace = std::getenv("DMI_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  int rc = dmi_mesh_prepare(mesh, cfg, &head, &job);
  if (rc) return rc;
  const double t_prep = ms();
  rc = dmi_job_encode(job, &att);
  const double t_enc = ms();
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
  if (trace) std::fprintf(stderr, "[dmi] encode_mesh: prepare %.1f ms, encode %.1f, job destroy %.1f, splice %.1f\n", t_prep, t_enc - t_prep, t_destroy - t_enc, ms() - t_destroy);
  return DMI_OK;
}

}  // extern "C"