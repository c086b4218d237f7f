// dmi_prepare_tables.cpp — the attribute corner tables of a connectivity group on the device (AttStage: dmi_conn.hip k_att_*) and the universal tables of a
// device-built group issued right behind its build (built_group_issue_tables).  Split out of dmi_prepare.cpp in round 5.
#include <atomic>
#include <thread>
#include "dmi_prepare.hpp"

using namespace dmi;

extern "C++" {
void dmi::AttStage::add(uint32_t member, uint32_t k, uint32_t F, uint32_t vcap, uint32_t map_off_words) {
  items.push_back({member, k, (uint32_t)corners, (uint32_t)verts, F});
  descs.push_back(AttItemDesc{member, map_off_words, (uint32_t)corners, (uint32_t)verts});
  corners += (size_t)F * 3; verts += vcap;
}
size_t dmi::AttStage::layout(size_t at) {
  if (corners >= (1ull << 32) || verts >= (1ull << 32)) { items.clear(); descs.clear(); corners = verts = 0; }   // (too large for one launch: the host builds them)
  rb_items = at; rb_info = rb_items + align256(items.size() * sizeof(AttItemDesc)); rb_seam = rb_info + align256(items.size() * sizeof(AttInfo));
  rb_c2v = rb_seam + align256(corners); rb_opp = rb_c2v + align256(corners * 4); rb_lmc = rb_opp + (want_opp ? align256(corners * 4) : 0);
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
  d_c2v = t.c2v; d_opp = t.opp; d_lmc = t.lmc; stream = s;
  lmc_first = std::min(corners, 2 * verts + 4096);
  HIP_TRY(hipMemcpyAsync(host + rb_info, t.info, items.size() * sizeof(AttInfo), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(host + rb_seam, t.seam, corners, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(host + rb_c2v, t.c2v, corners * 4, hipMemcpyDeviceToHost, s));
  if (want_opp) HIP_TRY(hipMemcpyAsync(host + rb_opp, t.opp, corners * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(host + rb_lmc, t.lmc, lmc_first * 4, hipMemcpyDeviceToHost, s));
  return DMI_OK;
}
int dmi::AttStage::complete() {
  if (items.empty() || !hp || !d_lmc) return DMI_OK;
  const AttInfo* info = reinterpret_cast<const AttInfo*>(hp + rb_info);
  size_t total = 0;
  for (size_t q = 0; q < items.size(); ++q) if (info[q].done) total = std::max(total, (size_t)info[q].pad + info[q].num_vertices);
  if (total <= lmc_first) return DMI_OK;
  if (total > corners) return fail(DMI_ERR_HIP, "attribute corner tables: vertex count out of range");
  HIP_TRY(hipMemcpyAsync(const_cast<uint8_t*>(hp) + rb_lmc + lmc_first * 4, d_lmc + lmc_first, (total - lmc_first) * 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  lmc_first = total;
  return DMI_OK;
}
}  // extern "C++"

void dmi::att_stage_fill(const AttStage& st, uint32_t member, uint32_t n_nonpos, std::vector<PrebuiltTable::Att>& out) {
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
    pa.opp = st.want_opp ? reinterpret_cast<const uint32_t*>(st.hp + st.rb_opp) + it->corner_off : nullptr;   // (null: seam ? none : the universal table's)
    pa.lmc = reinterpret_cast<const uint32_t*>(st.hp + st.rb_lmc) + info[q].pad;
    pa.d_c2v = st.d_c2v + it->corner_off; pa.d_opp = st.d_opp + it->corner_off;
  }
}

// The universal corner tables of every member of a device-built group: descriptors up, the dmi_conn.hip kernels, the tables back into the

int dmi::built_group_issue_tables(BuiltGroup& bg, hipStream_t s) {
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
  cn.mem.init(bg.device, s, C * 4 * (cn.any_mapped ? 5 : 4) + C + nv * 4 * 4 + nv + parts * 4 + (size_t)ND * (sizeof(ConnMeshDesc) + 8) + att_bytes + ((size_t)2 << 20));
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
  const bool host_att = dbg_on(DMI_DBG_HOST_ATT_TABLES);
  // (the map comparisons of all members side by side: two full-length memcmp per mesh with normals and texture coordinates — 90 MB for a 256-mesh stage —
  //  were 9 of a seam stage's 29 ms of build on the one thread that issues its tables)
  struct Pair { uint32_t mi; size_t a; };
  std::vector<Pair> pairs;
  for (uint32_t mi = 0; mi < ND && !host_att; ++mi) {
    const BuiltGroup::Member& mem = bg.members[mi];
    if (mem.atts.empty() || !mem.F) continue;
    for (size_t a = 0; a < mem.atts.size(); ++a)
      if (mem.atts[a].att_type != DMI_ATT_POSITION && mem.atts[a].map_off != (size_t)-1 && mem.atts[0].map_off != (size_t)-1) pairs.push_back({mi, a});
  }
  std::vector<uint8_t> maps_equal(pairs.size(), 0);
  {
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (size_t q; (q = next.fetch_add(1)) < pairs.size();) {
        const BuiltGroup::Member& mem = bg.members[pairs[q].mi];
        maps_equal[q] = std::memcmp(bg.h_a + mem.atts[pairs[q].a].map_off, bg.h_a + mem.atts[0].map_off, (size_t)mem.P * 4) == 0;
      }
    };
    const size_t n_threads = std::min<size_t>({pairs.size(), (size_t)host_threads(), (size_t)16});
    std::vector<dmi::Thread> th;
    for (size_t t = 1; t < n_threads; ++t) th.emplace_back(with_debug(work));
    work();
    for (auto& x : th) x.join();
  }
  size_t next_pair = 0;
  for (uint32_t mi = 0; mi < ND && !host_att; ++mi) {
    const BuiltGroup::Member& mem = bg.members[mi];
    if (mem.atts.empty() || !mem.F) continue;
    uint32_t k = 0;
    for (size_t a = 0; a < mem.atts.size(); ++a) {
      if (mem.atts[a].att_type == DMI_ATT_POSITION) continue;
      const size_t ma = mem.atts[a].map_off, mp = mem.atts[0].map_off;
      bool same = ma == (size_t)-1 && mp == (size_t)-1;
      if (ma != (size_t)-1 && mp != (size_t)-1) same = maps_equal[next_pair++] != 0;
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
  // Quad class (round 6): when no member has a point → value map — every attribute indexed like the positions, no attribute table of its own —
  // the host walks read the opposite corners as 4·face + k ids (no division by three per step: host_conn.cpp, what a whole-mesh call has done since round 5);
  // the device keeps its 3·face + k array for the relabelling and the fan rows, a converted copy goes down.
  cn.quad = false;
  if (C && st.items.empty() && !dbg_on(DMI_DBG_NO_QUAD | DMI_DBG_HOST_ATT_TABLES) && C < ((uint64_t)3 << 30)) {
    bool plain = true;
    for (uint32_t mi = 0; mi < ND && plain; ++mi) for (const auto& at : bg.members[mi].atts) if (at.map_off != (size_t)-1) { plain = false; break; }
    if (plain) {
      if (uint32_t* d_q = cn.mem.take<uint32_t>(C)) {
        launch_opp_quad(cn.d_opp, C, d_q, s);
        HIP_TRY(hipMemcpyAsync(hp + cn.rb_opp, d_q, C * 4, hipMemcpyDeviceToHost, s));
        cn.quad = true;
      }
    }
  }
  if (C && !cn.quad) HIP_TRY(hipMemcpyAsync(hp + cn.rb_opp, cn.d_opp, C * 4, hipMemcpyDeviceToHost, s));
  if (cn.any_mapped && C) HIP_TRY(hipMemcpyAsync(hp + cn.rb_c2v, cn.d_c2v, C * 4, hipMemcpyDeviceToHost, s));
  if (verts) HIP_TRY(hipMemcpyAsync(hp + cn.rb_lmc, d_lmc, (size_t)verts * 4, hipMemcpyDeviceToHost, s));
  if (verts) HIP_TRY(hipMemcpyAsync(hp + cn.rb_onb, d_onb, (size_t)verts, hipMemcpyDeviceToHost, s));
  if ((rc_att = st.issue(a, cn.mem, hp, s))) return rc_att;
  HIP_TRY(hipEventCreateWithFlags(&cn.ev, long_wait_flags()));
  HIP_TRY(hipEventRecord(cn.ev, s));
  cn.mem.pool.stream = nullptr; cn.mem.owner_waits = true;   // (the stream is a thread's library stream and the group may outlive the thread: its destructor waits for cn.ev instead)
  cn.issued = true;
  return DMI_OK;
}

// adopt (nullable): the groups are device-built ones (dmi_meshes_build) — which_all then lists their PRESENT members in group order
// (present[member] = index into the caller's arrays, -1 = not part of this call; the connectivity kernels run over every member: the
