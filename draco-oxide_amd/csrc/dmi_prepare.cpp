// dmi_prepare.cpp — whole-mesh entry points of the C ABI for ONE mesh: the connectivity stage (device tables + host walks), dmi_mesh_prepare,
// dmi_encode_mesh / dmi_encode_mesh_device, the device corner tables on their own, and the host-core stream coders on their own.  The batch forms
// are dmi_prepare_batch.cpp, the library streams dmi_streams.cpp (round 5 split).
#include "dmi_prepare.hpp"
#include <functional>

using namespace dmi;

namespace {

std::shared_ptr<StreamHolder> thread_stream(int device) { return library_thread_stream(device); }
using NumaPin = NumaScope;
// set by dmi_encode_mesh_device for the call it makes into mesh_prepare_impl: run once, right after the device stage has queued its last read-back
thread_local std::function<void(hipStream_t)> g_after_tables;

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
  // walks_only: the read-back feeds the library's own walks and nothing else — the opposite corners may come back as 4·face + k ids (see below)
  int build(const dmi_mesh* mesh, std::vector<uint32_t>& c2v_store, const uint32_t* src_faces = nullptr, const uint32_t* src_pos_map = nullptr, bool walks_only = false) {
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
    // The host's walks read `opposite` as 4·face + k ids when no attribute needs a corner table of its own — every attribute indexed exactly like the
    // Position attribute (the same map array, or none: per-point values) — so that nothing on the host reads the VALUES of the array but the two walks
    // (host_conn.cpp Enc4); the device keeps its own 3·face + k array for its kernels.
    bool quad = walks_only && F < (1u << 30) && !dbg_on(DMI_DBG_NO_QUAD);
    for (uint32_t i = 1; i < mesh->num_atts && quad; ++i) quad = mesh->atts[i].point_to_value == pos.point_to_value;
    mem.init(device, stream, C * 4 * (mapped ? 5 : 4) + (quad ? C * 4 : 0) + (mapped ? (size_t)P * 4 : 0) + nv * 4 * 4 + nv + C + parts * 4 + ((size_t)1 << 16));
    d_faces = src_faces ? const_cast<uint32_t*>(src_faces) : mem.take<uint32_t>(C);
    uint32_t* d_p2v = mapped ? (src_pos_map ? const_cast<uint32_t*>(src_pos_map) : mem.take<uint32_t>(P)) : nullptr;
    d_c2v = mapped ? mem.take<uint32_t>(C) : d_faces;
    d_opp = mem.take<uint32_t>(C);
    uint32_t* d_opp_q = quad ? mem.take<uint32_t>(C) : nullptr;
    if (quad && !d_opp_q) quad = false;
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
    if (quad) launch_opp_quad(d_opp, C, d_opp_q, stream);
    HIP_TRY(hipMemcpyAsync(hp_opp, quad ? d_opp_q : d_opp, C * 4, hipMemcpyDeviceToHost, stream));
    if (mapped) HIP_TRY(hipMemcpyAsync(hp_c2v, d_c2v, C * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(hp_onb, d_onb, Vcap, hipMemcpyDeviceToHost, stream));
    // (left-most corners: on the host only the construction of attribute tables with seams reads them — none exists for a mesh in the quad class, whose
    // walks take their boundary tests from the flags above: 4 bytes per vertex less on a stage bound by the link)
    if (!quad) HIP_TRY(hipMemcpyAsync(hp_lmc, d_lmc, (size_t)Vcap * 4, hipMemcpyDeviceToHost, stream));
    // (a whole-mesh call's early stage: issued behind the last read-back, so that its kernels run alone while the host walks — see dmi_encode_mesh_device)
    if (g_after_tables) { auto hook = std::move(g_after_tables); g_after_tables = nullptr; hook(stream); }
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
    pre.c2v = mapped ? hp_c2v : c2v_host; pre.opp = hp_opp; pre.lmc = quad ? nullptr : hp_lmc; pre.on_boundary = hp_onb; pre.V = V;
    pre.no_boundary = !(flags & CONN_HAS_BOUNDARY);
    pre.quad = quad;
    valid = true;
    return DMI_OK;
  }
};
void view_prebuilt(CornerTables& ct, const dmi_mesh* mesh, const PrebuiltTable& pre) {
  ct.F = mesh->num_faces; ct.V = pre.V;
  ct.c2p = mesh->faces; ct.c2v = pre.c2v; ct.opp = pre.opp; ct.lmc = pre.lmc;
  ct.no_boundary = pre.no_boundary;
  ct.quad = pre.quad;
  ct.att.clear();
}

}  // namespace

namespace dmi {
std::atomic<uint64_t> g_conn_us[4];
thread_local bool g_batch_worker_busy = false;
}  // namespace dmi

// dmi_encode_mesh_device's read-back of the faces / maps (on a stream of its own, beside the table kernels): waited for right before the host reads them
static thread_local hipEvent_t g_faces_event = nullptr;
// pre (nullable): the universal table already built by the device stage; view_faces: c2p may view the caller's face array (it outlives `o`)
extern "C++" int dmi::build_connectivity(const dmi_mesh* mesh, ConnOwner& o, std::vector<uint8_t>& bytes, const PrebuiltTable* pre, bool view_faces) {
  if (!mesh || !mesh->atts || mesh->num_atts == 0 || (!mesh->faces && mesh->num_faces)) return fail(DMI_ERR_INVALID_ARGUMENT, "bad mesh");
  if (mesh->atts[0].att_type != DMI_ATT_POSITION) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute 0 must be the Position attribute (core/mesh/builder.rs:115-125)");
  for (uint32_t i = 0; i < mesh->num_atts; ++i)
    if (mesh->atts[i].point_to_value == nullptr && mesh->atts[i].num_unique < mesh->atts[i].num_points) return fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + ": fewer values than points and no point_to_value map");
  std::string err;
  const bool trace = dbg_on(DMI_DBG_TRACE);
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
    // DMI_TEST_QUAD=1 (tests, host only): the walks over 4·face + k ids — what the device stage hands them for meshes none of whose attributes needs
    // a corner table of its own — on a table the HOST built: the bytes must not change
    if (dbg_on(DMI_DBG_TEST_QUAD) && o.ct.opp == o.ct.opp_own.data() && mesh->num_faces < (1u << 30)) {
      bool alike = true;
      for (uint32_t i = 1; i < mesh->num_atts && alike; ++i) alike = mesh->atts[i].point_to_value == mesh->atts[0].point_to_value;
      if (alike) {
        bool open_edge = false;
        for (uint32_t& v : o.ct.opp_own) { if (v != kNone) v += v / 3; else open_edge = true; }
        o.ct.quad = true;
        o.ct.no_boundary = !open_edge;   // (what the device stage reports: the walks of a closed mesh run without their tests for "none")
      }
    }
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
  std::shared_ptr<SeqStream> seq_stream = g_seq_stream;   // (a whole-mesh call ships the universal sequence to the device while the walk writes it)
  auto sequence_universal = [&] {
    TableRef tr{o.ct.F, o.ct.V, o.ct.c2v, o.ct.opp, o.ct.lmc, o.ct.quad, o.ct.no_boundary};
    attribute_sequence(tr, o.eb, o.seqs[0], boundary_flags ? boundary_flags : (on_boundary.empty() ? nullptr : on_boundary.data()), seq_stream ? &seq_stream->progress : nullptr);
    if (seq_stream) seq_stream->finish((uint32_t)o.seqs[0].size());
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
        pool_fit(a.opp, C);
        if (pa.opp) a.opp.assign(pa.opp, pa.opp + C);
        else {   // not read back (AttStage): the attribute's opposite corner is the universal one unless the edge is a seam of the attribute
          const uint32_t* uo = o.ct.opp;
          if (o.ct.quad) {
            a.opp.resize(C);
            uint32_t* ao = a.opp.data();
            for (size_t c = 0; c < C; ++c) { const uint32_t x = uo[c]; ao[c] = (pa.seam[c] || x == kNone) ? kNone : x - (x >> 2); }   // (4·face + k ids → 3·face + k)
          } else {
            // a plain copy, then the seam corners (few: eight flag bytes at a time) — the element-wise select after a zero-filling resize was 6 % of a seam transcode's CPU samples
            a.opp.assign(uo, uo + C);
            uint32_t* ao = a.opp.data();
            size_t c = 0;
            for (; c + 8 <= C; c += 8) {
              uint64_t w;
              std::memcpy(&w, pa.seam + c, 8);
              if (!w) continue;
              for (size_t k = 0; k < 8; ++k) if (pa.seam[c + k]) ao[c + k] = kNone;
            }
            for (; c < C; ++c) if (pa.seam[c]) ao[c] = kNone;
          }
        }
        pool_fit(a.lmc, pa.nv); a.lmc.assign(pa.lmc, pa.lmc + pa.nv);
        return;
      }
      o.ct.build_attribute_into(o.ct.att[k], maps[k], like_position(maps[k], map_points[k]));
    };
    if (maps.size() > 1 && overlap) {
      std::vector<dmi::Thread> th;
      for (size_t k = 0; k < maps.size(); ++k) th.emplace_back(with_debug(build_one), k);
      for (auto& x : th) x.join();
    } else {
      for (size_t k = 0; k < maps.size(); ++k) build_one(k);
    }
    for (size_t k = 0; k < maps.size(); ++k)
      for (size_t j = 0; j < k; ++j) if (maps[j] == maps[k]) { o.ct.copy_attribute_into(o.ct.att[k], o.ct.att[j]); o.ct.att[k].alias_of = (int)j; break; }
    t_att = since(a0);
  };
  dmi::Thread att_thread, seq_thread, flag_thread;
  EdgebreakerHooks hooks;
  if (overlap) {
    // the sequencer's per-vertex boundary test, ahead of time (beside the start of the traversal)
    if (!boundary_flags) flag_thread = dmi::Thread(with_debug([&] { TableRef tr{o.ct.F, o.ct.V, o.ct.c2v, o.ct.opp, o.ct.lmc}; vertex_boundary_flags(tr, on_boundary); }));
    att_thread = dmi::Thread(with_debug(build_att_tables));
    hooks.seeds_ready = [&] { if (flag_thread.joinable()) flag_thread.join(); seq_thread = dmi::Thread(with_debug([&] { const auto q0 = tick(); sequence_universal(); t_seq = since(q0); })); };
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
    std::vector<dmi::Thread> th;   // sequences of attribute tables with seams: independent walks
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
        if (overlap) th.emplace_back(with_debug(walk)); else walk();
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

extern "C" {

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
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
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
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
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
  st.want_opp = true;
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
  if ((rc = st.complete())) return rc;
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
static thread_local double g_tables_ms = 0;   // universal-table time of the running mesh_prepare_impl (device form)
int dmi_mesh_prepare(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** job) {
  return mesh_prepare_impl(mesh, cfg, header_and_connectivity, job, nullptr);
}
extern "C++" int dmi::mesh_prepare_impl(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** job, const DeviceMeshSrc* src) {
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
  if (!header_and_connectivity || !job) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  g_tables_ms = 0;
  NumaPin pin(cfg ? cfg->device : 0);
  const bool trace = dbg_on(DMI_DBG_TRACE);
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
                             (src || (!dbg_on(DMI_DBG_HOST_CONNECTIVITY) && mesh->num_faces >= (in_batch ? kDeviceRelabelMinFaces : kDeviceTablesMinFaces))) &&
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
      if ((rc = dt.build(mesh, o.ct.c2v_own, src ? src->faces : nullptr, src ? src->pos_map : nullptr, /*walks_only=*/true))) return rc;
      g_tables_ms = dt.t_down;
      if (trace) std::fprintf(stderr, "[dmi]   universal table of %u faces on the device: uploads issued %.2f ms, kernels + read-back issued %.2f, arrived %.2f (flags %#x)\n", mesh->num_faces, dt.t_up, dt.t_kernels, dt.t_down, dt.flags);
    }
    if (g_faces_event) { HIP_TRY(hipEventSynchronize(g_faces_event)); g_faces_event = nullptr; }   // (the host shadow of the faces / maps: in flight since before the table kernels)
    if (src && !dt.valid) return kNeedHostValues;
    rc = build_connectivity(mesh, o, bytes, dt.valid ? &dt.pre : nullptr, /*view_faces=*/true);
    if (rc) return rc;
    t_conn = ms();
    DeviceTableView view{dt.d_faces, dt.d_c2v, dt.d_opp, true, src != nullptr};
    if (g_one_shot_call && dt.valid && dt.stream) view.donor = &dt.mem.pool;   // (the job of a one-shot call may keep the stage's arrays: nothing of dt reads its memory after this)
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
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
  if (!out) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  NumaPin pin(cfg ? cfg->device : 0);   // (also around the encode: the host-core chains read the symbols the staging DMA brought in)
  dmi_buffer head{}, att{};
  dmi_job* job = nullptr;
  const bool trace = dbg_on(DMI_DBG_TRACE);
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  struct OneShot { OneShot() { g_one_shot_call = true; } ~OneShot() { g_one_shot_call = false; } } one_shot;
  int rc = dmi_mesh_prepare(mesh, cfg, &head, &job);
  if (rc) return rc;
  const double t_prep = ms();
  g_out_prefix = head.len;   // (the splice leaves room for header + connectivity: one buffer, no second copy of the streams)
  rc = dmi_job_encode(job, &att);
  g_out_prefix = 0;
  const double t_enc = ms();
  { const dmi_timings pre = g_last_call; g_last_call = job->last; g_last_call.tables_ms = pre.tables_ms; g_last_call.connectivity_ms = pre.connectivity_ms; g_last_call.job_create_ms = pre.job_create_ms; g_last_call.job_create_device_ms = pre.job_create_device_ms; }
  dmi_job_destroy(job);
  const double t_destroy = ms();
  if (rc) { dmi_free(&head); return rc; }
  // header + connectivity + attribute section in one library-owned buffer: the splice's own, with the front part filled in here
  std::memcpy(att.data, head.data, head.len);
  *out = att;
  dmi_free(&head);
  g_last_call.call_ms = (float)ms();
  if (trace) std::fprintf(stderr, "[dmi] encode_mesh: prepare %.1f ms, encode %.1f, job destroy %.1f, splice %.1f\n", t_prep, t_enc - t_prep, t_destroy - t_enc, ms() - t_destroy);
  return DMI_OK;
}

// encode::encode for a mesh whose buffers already live in HBM: `mesh` is a host struct whose faces, attribute values and point → value maps
// are DEVICE pointers on cfg->device.  Faces and maps are read back once (the Edgebreaker traversal and the sequencer are serial host walks),
// the values never leave the device; the tables are built from the device faces where they are.
int dmi_encode_mesh_device(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* out) {
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
  if (!out || !mesh || !mesh->atts || mesh->num_atts == 0 || mesh->num_atts > 255 || (!mesh->faces && mesh->num_faces)) return fail(DMI_ERR_INVALID_ARGUMENT, "bad mesh");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  const int device = cfg ? cfg->device : 0;
  HIP_TRY(hipSetDevice(device));
  NumaPin pin(device);
  auto holder = thread_stream(device);
  if (!holder) return fail(DMI_ERR_HIP, "hipStreamCreate");
  hipStream_t s = cfg && cfg->stream ? static_cast<hipStream_t>(cfg->stream) : holder->s;
  const bool trace = dbg_on(DMI_DBG_TRACE);
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
  // (on a stream of its own: the table kernels of mesh_prepare_impl run on the call's stream meanwhile; the host waits for both before its walks.  Round 6: also when the
  //  caller names a stream — the copy stream first waits for an event of that stream, so whatever the caller queued there to produce the faces is ordered before
  //  the read-back; on the caller's own stream the 2.2 ms copy of a 10M-triangle mesh ran IN FRONT of the table kernels instead of beside them)
  hipStream_t s_down = library_group_stream(device, 0);
  if (!s_down) s_down = s;
  if (s_down != s && cfg && cfg->stream) {
    hipEvent_t ready = nullptr;
    if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) == hipSuccess && hipEventRecord(ready, s) == hipSuccess && hipStreamWaitEvent(s_down, ready, 0) == hipSuccess) (void)hipEventDestroy(ready);
    else { (void)hipGetLastError(); if (ready) (void)hipEventDestroy(ready); s_down = s; }
  }
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
  struct OneShot { OneShot() { g_one_shot_call = true; } ~OneShot() { g_one_shot_call = false; g_after_tables = nullptr; g_early_quant.reset(); } } one_shot;
  // The values are in HBM and the device is about to idle through ≈ 90 ms of host walks (10M triangles): value ranges and the quantization in value
  // order run NOW, on the second group stream; the pass after the walks gathers packed values (EarlyQuant).  Large meshes only: a small mesh's encode
  // is bound by launches and runs its phase A from a captured graph.
  // Round 6: issued BEHIND the device stage's last read-back (an event of that stream): beside the table kernels and the 245 MB of copies the stage's
  // kernels ran at a fraction of their speed (k_value_ranges 523 µs on average in round 5's trace instead of 31), which no profile of the call could
  // tell from a slow kernel; now they run alone, a few ms into the host walks, and their hipEvent span is their kernel time.
  if (mesh->num_faces >= kDeviceTablesMinFaces) {
    dmi_config ec = cfg ? *cfg : dmi_config{};
    ec.device = device;
    hipStream_t s_early = library_group_stream(device, 1);
    if (s_early) g_after_tables = [mesh, ec, s_early](hipStream_t after) {
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
      if (hipEventRecord(ev, after) == hipSuccess && hipStreamWaitEvent(s_early, ev, 0) == hipSuccess) (void)early_quantize_issue(mesh->atts, mesh->num_atts, ec, s_early, g_early_quant);
      else (void)hipGetLastError();
      (void)hipEventDestroy(ev);
    };
  }
  // … and the universal sequence goes up WHILE the sequencer writes it (SeqStream): job creation finds it on the device
  if (mesh->num_faces >= kDeviceTablesMinFaces && !dbg_on(DMI_DBG_NO_EARLY | DMI_DBG_NO_SEQ_STREAM)) {
    auto ss = std::make_shared<SeqStream>();
    if (ss->start(device, library_group_stream(device, 0) != s ? library_group_stream(device, 0) : library_group_stream(device, 1), mesh->atts[0].num_unique)) g_seq_stream = ss;
  }
  struct SeqGuard { ~SeqGuard() { g_seq_stream.reset(); } } seq_guard;
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
  g_out_prefix = head.len;   // (see dmi_encode_mesh)
  rc = dmi_job_encode(job, &att);
  g_out_prefix = 0;
  { const dmi_timings pre = g_last_call; g_last_call = job->last; g_last_call.tables_ms = pre.tables_ms; g_last_call.connectivity_ms = pre.connectivity_ms; g_last_call.job_create_ms = pre.job_create_ms; g_last_call.job_create_device_ms = pre.job_create_device_ms; g_last_call.mesh_readback_ms = (float)t_down; }
  dmi_job_destroy(job);
  if (rc) { dmi_free(&head); return rc; }
  std::memcpy(att.data, head.data, head.len);
  *out = att;
  dmi_free(&head);
  g_last_call.call_ms = (float)ms();
  if (trace) std::fprintf(stderr, "[dmi] encode_mesh_device: faces + maps read back %.1f ms, prepare %.1f, encode + splice %.1f\n", t_down, t_prep - t_down, ms() - t_prep);
  return DMI_OK;
}

}  // extern "C"
