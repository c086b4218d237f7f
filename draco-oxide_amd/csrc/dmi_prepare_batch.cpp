// dmi_prepare_batch.cpp — the batch forms of the connectivity stage + job creation: dmi_meshes_prepare (host meshes), dmi_built_meshes_prepare (meshes
// dmi_meshes_build left on the device), dmi_shard_meshes / dmi_meshes_prepare_devices.  Split out of dmi_prepare.cpp in round 5 (the single-mesh entry
// points stayed there).
#include "dmi_prepare.hpp"

using namespace dmi;

extern "C" {

static int meshes_prepare_impl(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, const int32_t* device_of_mesh, dmi_buffer* header_and_connectivity, dmi_job** jobs);
int dmi_meshes_prepare(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** jobs) {
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
  return meshes_prepare_impl(meshes, n, cfg, nullptr, header_and_connectivity, jobs);
}
// One process, several GPUs: mesh j is prepared on HIP device device_of_mesh[j] (dmi_shard_meshes deals them by triangle count).
int dmi_meshes_prepare_devices(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, const int32_t* device_of_mesh, dmi_buffer* header_and_connectivity, dmi_job** jobs) {
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
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
  bool quad = false;   // the read-back `opp` holds 4·face + k ids (the walks' quad class)
  hipEvent_t ev_tables = nullptr, ev_values = nullptr;
  hipEvent_t ev_tables_borrowed = nullptr;   // an adopted group's tables were issued by its build: the event belongs to the BuiltGroup
  bool tables_in = false;
  std::mutex wait_mutex;
  std::atomic<int> issued{0};   // 1: phase 1 of this group is through (its fields are final, its tables on their way); -1: phase 1 failed — the walkers give up
  std::mutex issued_mutex;      // (the walkers of a group not yet issued sleep on issued_cv: sixteen of them polling every 50 µs were a tenth of a transcode's CPU samples)
  std::condition_variable issued_cv;
  void set_issued(int v) { { std::lock_guard<std::mutex> lock(issued_mutex); issued.store(v, std::memory_order_release); } issued_cv.notify_all(); }
  bool fail_unissued() { std::lock_guard<std::mutex> lock(issued_mutex); int z = 0; const bool changed = issued.compare_exchange_strong(z, -1); if (changed) issued_cv.notify_all(); return changed; }
  ~PrepGroup() {
    if (S) (void)hipStreamSynchronize(S);
    if (ev_tables) (void)hipEventDestroy(ev_tables);
    if (ev_values) (void)hipEventDestroy(ev_values);
    release_stage(stage);
  }
  int wait_tables() {   // (any worker: the first one blocks on the event, the others on the mutex)
    std::lock_guard<std::mutex> lock(wait_mutex);
    if (tables_in) return DMI_OK;
    HIP_TRY(long_wait_event(ev_tables_borrowed ? ev_tables_borrowed : ev_tables));
    if (int rc = (adopted ? adopted->conn.att : att).complete()) return rc;   // (left-most corners of the attribute tables beyond what came back with the stage: rare)
    tables_in = true;
    return DMI_OK;
  }
};
}  // namespace

struct AdoptedGroup { BuiltGroup* bg; std::vector<int32_t> present; };
static int prepare_slice_device(const dmi_mesh* meshes, const std::vector<uint32_t>& which_all, const dmi_config& cfg0, int device, uint32_t n_threads,
                                dmi_buffer* heads, dmi_job** jobs, std::vector<uint8_t>& done, const std::function<std::shared_ptr<StreamHolder>(uint32_t, int)>& worker_stream,
                                const std::vector<AdoptedGroup>* adopt = nullptr) {
  const uint32_t M = (uint32_t)which_all.size();
  if (!M) return DMI_OK;
  const bool trace = dbg_on(DMI_DBG_TRACE);
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  HIP_TRY(hipSetDevice(device));
  NumaScope pin(device);
  auto holder = library_thread_stream(device);
  if (!holder) return fail(DMI_ERR_HIP, "hipStreamCreate");
  hipStream_t S = holder->s;   // the coordinator's stream: job chunks are cleared on it, the deferred kernels of all jobs run on it
  struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } sync_on_exit{S};   // (also on error paths: a job the caller then destroys must not have its chunk cleared late)
  // ---- groups of ≈ 8M faces: the tables of the first arrive while the last is still being sent ----
  const uint64_t group_faces = dbg().prep_group_faces ? dbg().prep_group_faces : (uint64_t)(3u << 20);   // (measured, 256 meshes / 11M faces: 6M 21.5–22 ms, 3M 19.4–20.4, 1.5M 22.6–23.1)
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
    run_threads(nt, work);
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
    if (g.issued.load(std::memory_order_acquire) != 1) {   // (the coordinator is still packing / sending this group)
      std::unique_lock<std::mutex> lock(g.issued_mutex);
      g.issued_cv.wait(lock, [&] { return g.issued.load(std::memory_order_acquire) != 0; });
      if (g.issued.load(std::memory_order_acquire) < 0) return DMI_ERR_HIP;
    }
    if (!g.adopted) {   // this mesh's values → staging → device, behind the group's table kernels on its stream
      size_t lo = (size_t)-1, hi = 0;
      for (uint32_t a = 0; a < m.num_atts; ++a) {
        const dmi_attribute& at = m.atts[a];
        const size_t vb = (size_t)at.num_unique * at.num_components * 4;
        if (!vb) continue;
        stream_copy(g.hp + g.lay[k].values[a], at.values, vb);
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
    if (g.adopted ? g.adopted->conn.quad : g.quad) { pre.quad = true; pre.lmc = nullptr; }   // (4·face + k ids in `opp`: built_group_issue_tables)
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
    const bool defer_seams = !dbg_on(DMI_DBG_NO_DEFER_SEAMS);
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
  dmi::Thread walkers;
  int rc_walk = DMI_OK;
  std::string err_walk;
  struct JoinWalkers { dmi::Thread& t; std::vector<std::unique_ptr<PrepGroup>>& gs; ~JoinWalkers() { if (t.joinable()) { for (auto& g : gs) (void)g->fail_unissued(); t.join(); } } } join_walkers{walkers, groups};
  auto pack_threads = [&]() -> uint32_t { return walkers.joinable() ? std::max(2u, n_threads / 4) : 0u; };   // (0 = all: nothing else runs yet)
  // ---- phase 1, group by group: layout, pack, send, build the tables, fetch them (nothing here waits for the device) ----
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    PrepGroup& g = *groups[gi];
    const uint32_t Mg = (uint32_t)g.which.size();
    g.S = library_group_stream(device, (int)(gi & 1));
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
      g.set_issued(1);
      if (!walkers.joinable() && M > 1) walkers = dmi::Thread(with_debug([&] { rc_walk = parallel_over(M, walk_one, nullptr); if (rc_walk) err_walk = g_last_error; }));
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
      const bool host_att = dbg_on(DMI_DBG_HOST_ATT_TABLES);
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
    g.mem.init(device, g.S, up_bytes + C * 4 * (g.any_mapped ? 5 : 4) + C + nv * 4 * 4 + nv + parts * 4 + (size_t)Mg * (sizeof(ConnMeshDesc) + 8) + g.att.device_bytes() + ((size_t)1 << 20));
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
          stream_copy(hp + g.lay[k].faces, m.faces, (size_t)m.num_faces * 12);
          if (g.lay[k].mapped) stream_copy(hp + g.lay[k].pos_map, m.atts[0].point_to_value, (size_t)m.atts[0].num_points * 4);
          for (uint32_t i = 1; i < m.num_atts; ++i) {
            const dmi_attribute& at = m.atts[i];
            if (!at.point_to_value || at.point_to_value == m.atts[0].point_to_value) continue;
            bool first = true;
            for (uint32_t j = 1; j < i; ++j) if (m.atts[j].point_to_value == at.point_to_value) first = false;
            if (first) stream_copy(hp + g.lay[k].maps[i], at.point_to_value, (size_t)at.num_points * 4);
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
    // quad class (as built_group_issue_tables): every attribute of every mesh of the group indexed by the position attribute's map array (or by none) — the walks
    // read 4·face + k ids; the device keeps its 3·face + k array, a converted copy goes down
    g.quad = false;
    if (C && g.att.items.empty() && !dbg_on(DMI_DBG_NO_QUAD | DMI_DBG_HOST_ATT_TABLES) && C < ((uint64_t)3 << 30)) {
      bool plain = true;
      for (uint32_t k = 0; k < Mg && plain; ++k) { const dmi_mesh& m = meshes[g.which[k]]; for (uint32_t i = 1; i < m.num_atts; ++i) if (m.atts[i].point_to_value != m.atts[0].point_to_value) { plain = false; break; } }
      if (plain) {
        if (uint32_t* d_q = g.mem.take<uint32_t>(C)) {
          launch_opp_quad(g.d_opp, C, d_q, g.S);
          HIP_TRY(hipMemcpyAsync(hp + g.rb_opp, d_q, C * 4, hipMemcpyDeviceToHost, g.S));
          g.quad = true;
        }
      }
    }
    if (!g.quad) HIP_TRY(hipMemcpyAsync(hp + g.rb_opp, g.d_opp, C * 4, hipMemcpyDeviceToHost, g.S));
    if (g.any_mapped) HIP_TRY(hipMemcpyAsync(hp + g.rb_c2v, g.d_c2v, C * 4, hipMemcpyDeviceToHost, g.S));
    if (want_lmc) HIP_TRY(hipMemcpyAsync(hp + g.rb_lmc, d_lmc, (size_t)g.total_verts * 4, hipMemcpyDeviceToHost, g.S));
    HIP_TRY(hipMemcpyAsync(hp + g.rb_onb, d_onb, (size_t)g.total_verts, hipMemcpyDeviceToHost, g.S));
    if (!g.att.items.empty()) { want_lmc = true; HIP_TRY(hipMemcpyAsync(hp + g.rb_lmc, d_lmc, (size_t)g.total_verts * 4, hipMemcpyDeviceToHost, g.S)); }
    if ((rc = g.att.issue(a, g.mem, hp, g.S))) return rc;
    HIP_TRY(hipEventCreateWithFlags(&g.ev_tables, long_wait_flags()));
    HIP_TRY(hipEventRecord(g.ev_tables, g.S));
    // part B (the values — more than half of the bytes) is packed and sent mesh by mesh by the walkers, first thing, while they would otherwise
    // wait for this group's tables (walk_one): phase 1 — what every walker waits for — packs faces and maps only
    HIP_TRY(hipEventCreateWithFlags(&g.ev_values, hipEventDisableTiming));
    g.set_issued(1);
    if (!walkers.joinable() && M > 1) walkers = dmi::Thread(with_debug([&] { rc_walk = parallel_over(M, walk_one, nullptr); if (rc_walk) err_walk = g_last_error; }));
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
    b.any_sorted = 0;
    for (const RelabelItem& it : items) if (!it.plain) b.any_sorted = 1;
    b.total_faces = (uint32_t)rf; b.total_verts = (uint32_t)rv; b.total_keys = (uint32_t)rk; b.total_seq = (uint32_t)rs; b.total_remap_faces = (uint32_t)rr;
    launch_clear_items(reinterpret_cast<const CopyItem*>(d2 + off_clears), (uint32_t)clears.size(), S);
    launch_scatter_items(reinterpret_cast<const CopyItem*>(d2 + off_copies), (uint32_t)copies.size(), base, S);
    launch_relabel_batch(b, S);
    launch_compose_batch(reinterpret_cast<const ComposeItem*>(d2 + off_comps), (uint32_t)comps.size(), (uint32_t)comp_total, S);
    launch_build_fans_batch(reinterpret_cast<const FanItem*>(d2 + off_fans), (uint32_t)fans.size(), (uint32_t)fan_total, S);
  }
  // (while the device works: the meshes' host tables go back to the pool — a thousand small frees — on the worker threads)
  (void)parallel_over(M, [&](uint32_t, uint32_t kk) -> int { owners[kk].reset(); return DMI_OK; }, nullptr);
  HIP_TRY(long_wait_stream(S));
  const double t_dev = ms();
  groups.clear();
  if (trace || dbg_on(DMI_DBG_TRACE_STAGES)) {
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
  if (!dbg_on(DMI_DBG_HOST_CONNECTIVITY) && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
    std::vector<int> devices;
    for (uint32_t j = 0; j < n; ++j) { const int d = device_of_mesh ? device_of_mesh[j] : (cfg ? cfg->device : 0); if (std::find(devices.begin(), devices.end(), d) == devices.end()) devices.push_back(d); }
    const uint32_t min_faces = dbg().batch_min_faces ? (uint32_t)dbg().batch_min_faces : 1u;
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
    else { std::vector<dmi::Thread> th; for (size_t g = 0; g < devices.size(); ++g) th.emplace_back(with_debug(run_device), g); for (auto& x : th) x.join(); }
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
  else run_threads(n_threads, work);
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
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
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
    if (large || dbg_on(DMI_DBG_HOST_CONNECTIVITY)) { for (int32_t j : ag.present) if (j >= 0) singles.push_back((uint32_t)j); continue; }
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
    run_threads(nt, work);
    for (size_t k = 0; k < singles.size(); ++k) if (rcs[k]) { g_last_error = errs[k]; return bail(rcs[k]); }
  }
  return DMI_OK;
}

}  // extern "C"
