// dmi_build.cpp — dmi_meshes_build: MeshBuilder::build (core/mesh/builder.rs:62-90) + Attribute::from (core/attribute/mod.rs:394-452) for a
// batch of glTF primitives (io/gltf/decode.rs:2328-2525 builds one Mesh per primitive) on the device — SURVEY §8f-2.
//
// Host side of dmi_build.hip.  The primitives are packed into groups of ≈ 6M faces (the groups dmi_built_meshes_prepare's connectivity stage
// then takes over as they are); per group: the accessors' rows and the indices are copied into pinned staging by the library's host
// threads (strided accessors are de-strided on the way), ONE copy up, the build kernels (one launch per kernel for all primitives of the
// group), a small read-back of counts and offsets, then ONE read-back of arena A (faces + point → value maps: what the host's serial walks
// read) — the unique values stay in arena B on the device unless the caller asks for them.  Two streams alternate between consecutive
// groups, so a group's kernels and read-back overlap the next group's packing and upload.  IN PLACE (round 5): a group whose accessors and index
// arrays all lie in memory the importer got from dmi_host_alloc (page-locked blocks of the library's: dmi_hostmem.cpp) is not packed at all — the
// DMA engine copies the caller's bytes as they lie (the arrays' ranges merged into spans: about one copy per file), the kernels read the rows with
// their accessor's stride, one launch gathers the index arrays (u8 / u16 widened) into the face array.  (A kernel reading host memory directly was
// tried first: it moves the bytes at the link's rate too, but it does so on compute units — 29 ms of device time per 1024-file transcode beside
// the kernels of three other stages — where the copy engines are idle.)  A primitive the device form does not cover
// (see dmi_build.hip) is built by the host builder (dmi_mesh_build) inside the same call: same result either way
// (tests/test_gpu_device_build.py holds every mesh equal to dmi_mesh_build's and to the oracle's restated builder).
#include "dmi_job.hpp"

using namespace dmi;

namespace dmi { bool host_in_place(const void* p, size_t bytes); }   // dmi_hostmem.cpp: inside a dmi_host_alloc block?

namespace {

thread_local dmi_build_timings g_last_build{};

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
size_t component_bytes(uint8_t t) {
  switch (t) {
    case DMI_U8: case DMI_I8: return 1;
    case DMI_U16: case DMI_I16: return 2;
    case DMI_U32: case DMI_I32: case DMI_F32: return 4;
    case DMI_U64: case DMI_I64: case DMI_F64: return 8;
    default: return 0;
  }
}
uint32_t pow2_at_least(uint64_t n) { uint32_t p = 16; while (p < n) p <<= 1; return p; }

// fn(worker, i) for i in [0, count) on up to n_threads host threads, heaviest first when `weight` is given
int run_parallel(uint32_t count, uint32_t n_threads, const std::function<int(uint32_t)>& fn, const std::function<uint64_t(uint32_t)>& weight) {
  if (!count) return DMI_OK;
  std::vector<uint32_t> order(count);
  for (uint32_t k = 0; k < count; ++k) order[k] = k;
  if (weight) std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return weight(x) > weight(y); });
  std::atomic<uint32_t> next{0};
  const uint32_t nt = std::max(1u, std::min(n_threads, count));
  std::vector<int> rcs(nt, DMI_OK);
  std::vector<std::string> errs(nt);
  auto work = [&](uint32_t t) {
    for (uint32_t i; (i = next.fetch_add(1)) < count;) { const int rc = fn(order[i]); if (rc) { rcs[t] = rc; errs[t] = g_last_error; next.store(count); return; } }
  };
  run_threads(nt, work);
  for (uint32_t t = 0; t < nt; ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
  return DMI_OK;
}

// rows of one accessor, tightly packed, into dst (bytes [lo, hi) of the packed form: large accessors are copied in slices)
void pack_rows(uint8_t* dst, const dmi_raw_accessor& a, size_t row_bytes, size_t lo, size_t hi) {
  const uint8_t* src = static_cast<const uint8_t*>(a.data);
  if (!a.byte_stride || a.byte_stride == row_bytes) { if (dbg_on(DMI_DBG_NO_STREAM_COPY)) std::memcpy(dst + lo, src + lo, hi - lo); else stream_copy(dst + lo, src + lo, hi - lo); return; }
  for (size_t r = lo / row_bytes, e = (hi + row_bytes - 1) / row_bytes; r < e; ++r) std::memcpy(dst + r * row_bytes, src + r * (size_t)a.byte_stride, row_bytes);
}

// can the kernels take this primitive?
bool device_form(const dmi_raw_mesh& m) {
  if (!m.n_atts || m.n_atts > kMbMaxAtts || !m.num_faces || !m.indices) return false;
  if (m.index_type != DMI_U8 && m.index_type != DMI_U16 && m.index_type != DMI_U32) return false;
  const uint32_t P = m.atts[0].count;
  if (!P || P >= (1u << 30) || m.num_faces >= (1u << 30)) return false;
  // (the kernels index a group's hash tables, rows and arenas with 32-bit word offsets: a primitive whose own tables pass 2^31 words — about 300M
  // points with four attributes — is the host builder's, whatever else is in the call)
  uint64_t tab_words = 0, row_words = 0;
  for (uint32_t i = 0; i < m.n_atts; ++i) { tab_words += pow2_at_least(2ull * P); row_words += (uint64_t)P * m.atts[i].num_components + 64; }
  if (tab_words >= (1ull << 31) || row_words + 3ull * m.num_faces >= (1ull << 31) || (uint64_t)P * m.n_atts >= (1ull << 30)) return false;
  for (uint32_t i = 0; i < m.n_atts; ++i) {
    const dmi_raw_accessor& a = m.atts[i];
    if (a.count != P || component_bytes(a.component_type) != 4 || a.num_components < 1 || a.num_components > 4 || !a.data) return false;
    if (a.byte_stride && a.byte_stride < 4u * a.num_components) return false;
  }
  return true;
}

// MeshBuilder::dependency_check (builder.rs:95-111) and the argument checks dmi_mesh_build makes, for every primitive
int check_raw(const dmi_raw_mesh& m, uint32_t j) {
  const std::string who = "primitive " + std::to_string(j) + ": ";
  if ((!m.atts && m.n_atts) || (!m.indices && m.num_faces)) return fail(DMI_ERR_INVALID_ARGUMENT, who + "null");
  for (uint32_t i = 0; i < m.n_atts; ++i) {
    const dmi_raw_accessor& a = m.atts[i];
    if (!component_bytes(a.component_type) || a.num_components == 0) return fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, who + "attribute " + std::to_string(i) + ": bad component type / count");
    if (!a.data && a.count) return fail(DMI_ERR_INVALID_ARGUMENT, who + "attribute " + std::to_string(i) + ": no data");
    for (uint32_t k = 0; k < a.num_parents; ++k) if (a.parents[k] >= m.n_atts) return fail(DMI_ERR_BAD_PARENT, who + "attribute " + std::to_string(i) + ": parent id out of range");
    if (a.att_type == DMI_ATT_TEXCOORD) {
      bool ok = false;
      for (uint32_t k = 0; k < a.num_parents; ++k) if (m.atts[a.parents[k]].att_type == DMI_ATT_POSITION) ok = true;
      if (!ok) return fail(DMI_ERR_BAD_PARENT, who + "MinimumDependencyError(TextureCoordinate, Position)");
    }
  }
  if (m.num_faces && m.index_type != DMI_U8 && m.index_type != DMI_U16 && m.index_type != DMI_U32) return fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, who + "index type must be U8 / U16 / U32");
  return DMI_OK;
}

// the host builder for one primitive (rows de-strided, indices widened first)
int build_on_host(const dmi_raw_mesh& m, dmi_built_mesh* out) {
  std::vector<std::vector<uint8_t>> rows(m.n_atts);
  std::vector<dmi_raw_attribute> atts(m.n_atts);
  for (uint32_t i = 0; i < m.n_atts; ++i) {
    const dmi_raw_accessor& a = m.atts[i];
    const size_t rb = component_bytes(a.component_type) * a.num_components;
    const void* data = a.data;
    if (a.byte_stride && a.byte_stride != rb) {
      rows[i].resize((size_t)a.count * rb);
      pack_rows(rows[i].data(), a, rb, 0, rows[i].size());
      data = rows[i].data();
    }
    atts[i] = dmi_raw_attribute{data, a.count, a.component_type, a.num_components, a.att_type, a.domain, a.num_parents, a.parents};
  }
  std::vector<uint32_t> wide;
  const uint32_t* faces = static_cast<const uint32_t*>(m.indices);
  if (m.index_type != DMI_U32 && m.num_faces) {
    wide.resize((size_t)m.num_faces * 3);
    if (m.index_type == DMI_U8) { const uint8_t* s = static_cast<const uint8_t*>(m.indices); for (size_t k = 0; k < wide.size(); ++k) wide[k] = s[k]; }
    else { const uint16_t* s = static_cast<const uint16_t*>(m.indices); for (size_t k = 0; k < wide.size(); ++k) wide[k] = s[k]; }
    faces = wide.data();
  }
  return dmi_mesh_build(atts.data(), m.n_atts, faces, m.num_faces, out);
}

// One group of primitives on its way through the device
struct BuildGroup {
  std::vector<uint32_t> which;          // indices into the caller's array
  uint64_t raw_faces = 0, points = 0, ap = 0, tab_words = 0, row_words = 0;   // (the last two: what the 32-bit word offsets of the kernels must hold)
  hipStream_t S = nullptr;
  TempDev scratch;
  HostStage* up_stage = nullptr;
  std::shared_ptr<BuiltGroup> built;
  std::vector<MbMesh> meshes;
  std::vector<MbItem> items;
  std::vector<size_t> row_at;           // per item: byte offset of its packed rows in the upload region
  std::vector<size_t> idx_at;           // per mesh: byte offset of its indices as uploaded
  size_t up_bytes = 0, values_bytes = 0;
  size_t a_cap_words = 0, b_cap_words = 0;
  // read-back of the counts (pinned, tail of the keep stage)
  MbMeshOut* h_mesh_out = nullptr; MbItemOut* h_item_out = nullptr; uint32_t* h_totals = nullptr;
  MbArgs args{};
  hipEvent_t ev_counts = nullptr, ev_done = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
  bool host_values = false, large = false;
  bool in_place = false;                // every array of every member lies in page-locked host memory: DMA straight out of the caller's buffers, no pack
  struct Span { uintptr_t lo, hi; size_t dev_off; };
  std::vector<Span> spans;              // in place: merged host ranges, one DMA each
  bool settled = false;                 // ev_done has been waited for: nothing of the build is in flight (work issued on S AFTER it — the group's connectivity stage — uses memory of its own)
  ~BuildGroup() {
    if (S && !settled) (void)hipStreamSynchronize(S);
    for (hipEvent_t e : {ev_counts, ev_done, ev_k0, ev_k1}) if (e) (void)hipEventDestroy(e);
    release_stage(up_stage);
  }
};

}  // namespace

extern "C" {

int dmi_last_build_timings(dmi_build_timings* t) {
  if (!t) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  *t = g_last_build;
  return DMI_OK;
}

// face / point counts of n built meshes in one call, and their release in one call (a transcode driver in an interpreted language pays per call)
int dmi_built_meshes_info(const dmi_built_mesh* built, uint32_t n, uint32_t* num_faces, uint32_t* num_points) {
  if ((!built && n) || !num_faces || !num_points) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  for (uint32_t j = 0; j < n; ++j) {
    num_faces[j] = built[j].mesh.num_faces;
    num_points[j] = built[j].mesh.num_atts && built[j].mesh.atts ? built[j].mesh.atts[0].num_points : 0u;
  }
  return DMI_OK;
}
void dmi_built_meshes_free(dmi_built_mesh* built, uint32_t n) {
  if (!built) return;
  for (uint32_t j = 0; j < n; ++j) dmi_built_mesh_free(&built[j]);
}

int dmi_meshes_build(const dmi_raw_mesh* raw, uint32_t n, const dmi_config* cfg, uint32_t flags, dmi_built_mesh* out) {
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
  if (!raw || !out || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  for (uint32_t j = 0; j < n; ++j) out[j] = dmi_built_mesh{};
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  const int device = cfg ? cfg->device : 0;
  if (device < 0 || device >= ndev) return fail(DMI_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  NumaScope pin(device);
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  const bool trace = dbg_on(DMI_DBG_TRACE);
  const bool host_values = (flags & DMI_BUILD_HOST_VALUES) != 0;
  const uint32_t n_threads = std::max(1u, std::min((uint32_t)host_threads(), kMaxPrepareWorkers));
  const uint32_t pack_threads = std::max(2u, n_threads / 4);   // (copies into staging: a few cores saturate the memory bus; a transcode pipeline runs the previous stage's host walks beside this)
  int rc;
  for (uint32_t j = 0; j < n; ++j) if ((rc = check_raw(raw[j], j))) return rc;
  struct Cleanup { dmi_built_mesh* out; uint32_t n; bool armed = true; ~Cleanup() { if (armed) for (uint32_t j = 0; j < n; ++j) dmi_built_mesh_free(&out[j]); } } cleanup{out, n};

  // ---- groups of ≈ 6M faces (dmi_built_meshes_prepare's connectivity stage takes a group as it is); a primitive of 2^20 faces or more
  //      is a group of its own (it goes through the single-mesh prepare) ----
  const uint64_t group_faces = dbg().prep_group_faces ? dbg().prep_group_faces : (uint64_t)(6u << 20);
  std::vector<std::unique_ptr<BuildGroup>> groups;
  std::vector<uint32_t> host_list;
  const bool no_device = dbg_on(DMI_DBG_HOST_BUILD);
  const bool no_ingest = dbg_on(DMI_DBG_NO_IN_PLACE);   // (A/B: pack + copy even what could go up where it lies)
  for (uint32_t j = 0; j < n; ++j) {
    const dmi_raw_mesh& m = raw[j];
    if (no_device || !device_form(m)) { host_list.push_back(j); continue; }
    const bool large = m.num_faces >= kDeviceRelabelMinFaces;
    const uint64_t ap = (uint64_t)m.atts[0].count * m.n_atts;
    // can the DMA engine take this primitive's arrays where they lie?
    bool in_place = !no_ingest;
    for (uint32_t i = 0; in_place && i <= m.n_atts; ++i) {
      const void* p; size_t span; bool aligned;
      if (i < m.n_atts) {
        const dmi_raw_accessor& a = m.atts[i];
        const size_t row = 4u * a.num_components, stride = a.byte_stride ? a.byte_stride : row;
        p = a.data; span = (size_t)(a.count - 1) * stride + row; aligned = !((uintptr_t)p & 3) && !(stride & 3);
      } else {
        const size_t eb = component_bytes(m.index_type);
        p = m.indices; span = (size_t)m.num_faces * 3 * eb; aligned = !((uintptr_t)p & (eb - 1));
      }
      in_place = aligned && host_in_place(p, span);
    }
    bool fresh = groups.empty() || large || groups.back()->large || groups.back()->in_place != in_place;
    if (!fresh) {
      const BuildGroup& g = *groups.back();
      uint64_t tab = 0, rows = 0;
      for (uint32_t i = 0; i < m.n_atts; ++i) { tab += pow2_at_least(2ull * m.atts[0].count); rows += (uint64_t)m.atts[0].count * m.atts[i].num_components + 64; }
      fresh = g.raw_faces + m.num_faces > group_faces || g.ap + ap >= (1ull << 30) || g.which.size() >= 65536 ||
              g.tab_words + tab >= (1ull << 32) || g.row_words + rows + 3ull * (g.raw_faces + m.num_faces) >= (1ull << 32);
    }
    if (fresh) groups.emplace_back(new BuildGroup());
    BuildGroup& g = *groups.back();
    g.large = large;
    g.in_place = in_place;
    MbMesh me{};
    me.index = (uint32_t)g.meshes.size(); me.n_items = m.n_atts; me.item0 = (uint32_t)g.items.size(); me.P = m.atts[0].count; me.F = m.num_faces;
    me.face_off = (uint32_t)g.raw_faces; me.point_off = (uint32_t)g.points;
    for (uint32_t i = 0; i < m.n_atts; ++i) {
      MbItem it{};
      it.mesh = me.index; it.P = me.P; it.words = m.atts[i].num_components; it.is_float = m.atts[i].component_type == DMI_F32;
      it.stride = it.words;
      it.ap_off = (uint32_t)(g.ap + (uint64_t)i * me.P);
      g.items.push_back(it);
    }
    g.which.push_back(j);
    g.meshes.push_back(me);
    g.raw_faces += m.num_faces; g.points += me.P; g.ap += ap;
    for (uint32_t i = 0; i < m.n_atts; ++i) { g.tab_words += pow2_at_least(2ull * me.P); g.row_words += (uint64_t)me.P * m.atts[i].num_components + 64; }
  }
  double t_pack = 0;
  uint64_t bytes_up = 0, bytes_down = 0;

  // ---- issue: layout, pack, upload, kernels, counts read-back (nothing here waits for the device) ----
  auto issue = [&](size_t gi) -> int {
    BuildGroup& g = *groups[gi];
    const uint32_t M = (uint32_t)g.meshes.size(), NI = (uint32_t)g.items.size();
    g.host_values = host_values;
    g.S = library_group_stream(device, (int)(gi & 1));
    if (!g.S) return fail(DMI_ERR_HIP, "hipStreamCreate");
    // upload region: packed rows of every item (256-byte aligned), then the indices (u32 as they are; narrower ones are widened on the device).
    // In place: the caller's bytes as they lie — the arrays' host ranges merged into spans (arrays of one file sit next to each other: ONE DMA per
    // file instead of one per accessor — 4096 accessor-sized copies take 3 × the time of the bytes they move), a span's device copy at the same
    // address modulo 256; the kernels read rows with their accessor's stride, one launch gathers the index arrays into the face array.
    size_t at = 0, vtab_words = 0, ptab_words = 0;
    g.row_at.resize(NI); g.idx_at.resize(M);
    if (g.in_place) {
      struct Arr { uintptr_t lo, hi; uint32_t k, i; };
      std::vector<Arr> arrs;
      for (uint32_t k = 0; k < M; ++k) {
        const dmi_raw_mesh& m = raw[g.which[k]];
        for (uint32_t i = 0; i < m.n_atts; ++i) {
          const dmi_raw_accessor& ac = m.atts[i];
          const size_t row = 4u * ac.num_components, stride = ac.byte_stride ? ac.byte_stride : row;
          arrs.push_back({(uintptr_t)ac.data, (uintptr_t)ac.data + (size_t)(ac.count - 1) * stride + row, k, i});
        }
        arrs.push_back({(uintptr_t)m.indices, (uintptr_t)m.indices + (size_t)m.num_faces * 3 * component_bytes(m.index_type), k, m.n_atts});
      }
      std::vector<uint32_t> order(arrs.size());
      for (uint32_t q = 0; q < order.size(); ++q) order[q] = q;
      std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return arrs[x].lo < arrs[y].lo; });
      constexpr uintptr_t kGap = 16384;   // bytes of no use that ride along rather than start a new copy
      for (uint32_t q : order) {
        const Arr& ar = arrs[q];
        if (g.spans.empty() || ar.lo > g.spans.back().hi + kGap || !host_in_place(reinterpret_cast<const void*>(g.spans.back().lo), std::max(ar.hi, g.spans.back().hi) - g.spans.back().lo)) {
          at = align256(at) + (ar.lo & 255);
          g.spans.push_back({ar.lo, ar.hi, at});
        } else {
          g.spans.back().hi = std::max(g.spans.back().hi, ar.hi);
        }
        at = g.spans.back().dev_off + (g.spans.back().hi - g.spans.back().lo);
        const size_t dev = g.spans.back().dev_off + (ar.lo - g.spans.back().lo);
        if (ar.i < raw[g.which[ar.k]].n_atts) g.row_at[g.meshes[ar.k].item0 + ar.i] = dev; else g.idx_at[ar.k] = dev;
      }
      at = align256(at);
      //  the spans carry whatever lies between the arrays — interleaved strides, unused attributes, gaps of up to 16 KiB, accessors shared
      // between primitives — so their total is unrelated to the packed estimate the group was formed with: a group whose spans come to several times its
      // packed size, or near the kernels' 2^34-byte reach, is packed like any other instead of failing the call
      size_t packed = 0;
      for (uint32_t k = 0; k < M; ++k) {
        const dmi_raw_mesh& m = raw[g.which[k]];
        for (uint32_t i = 0; i < m.n_atts; ++i) packed += (size_t)m.atts[i].count * m.atts[i].num_components * 4;
        packed += (size_t)m.num_faces * 3 * component_bytes(m.index_type);
      }
      if (at >= ((size_t)1 << 33) || at > 3 * packed + ((size_t)64 << 20)) { g.in_place = false; g.spans.clear(); at = 0; }
    }
    for (uint32_t i = 0; i < NI; ++i) {
      MbItem& it = g.items[i];
      if (!g.in_place) { g.row_at[i] = at; at = align256(at + (size_t)it.P * it.words * 4); }
      it.row_off = (uint32_t)(g.row_at[i] / 4);
      const uint32_t ts = pow2_at_least(2ull * it.P);
      it.tab_off = (uint32_t)vtab_words; it.tab_mask = ts - 1;
      vtab_words += ts;
    }
    if (g.in_place) for (uint32_t k = 0; k < M; ++k) {
      const dmi_raw_mesh& m = raw[g.which[k]];
      for (uint32_t i = 0; i < m.n_atts; ++i) if (m.atts[i].byte_stride) g.items[g.meshes[k].item0 + i].stride = m.atts[i].byte_stride / 4;
    }
    g.values_bytes = at;
    if (at >= ((size_t)1 << 34) || vtab_words >= (1ull << 32)) return fail(DMI_ERR_INVALID_ARGUMENT, "build group too large");
    bool any_narrow = g.in_place;   // (in place: the face array is always gathered out of the spans)
    if (!g.in_place) for (uint32_t k = 0; k < M; ++k) if (component_bytes(raw[g.which[k]].index_type) != 4) any_narrow = true;
    for (uint32_t k = 0; k < M; ++k) {
      const dmi_raw_mesh& m = raw[g.which[k]];
      MbMesh& me = g.meshes[k];
      const uint32_t ts = pow2_at_least(2ull * me.P);
      me.ptab_off = (uint32_t)ptab_words; me.ptab_mask = ts - 1;
      ptab_words += ts;
      if (g.in_place) continue;
      // indices: a group of 32-bit index arrays uploads them as ONE array (the kernels' face array); with a narrower array in the group
      // every array goes up as it is and one launch widens / copies them into the face array
      g.idx_at[k] = at;
      at += (size_t)m.num_faces * 3 * component_bytes(m.index_type);
      if (any_narrow) at = align256(at);
    }
    at = align256(at);
    g.up_bytes = at;
    const size_t RF = (size_t)g.raw_faces, PT = (size_t)g.points, AP = (size_t)g.ap;
    // arena capacities (words): A = faces | maps, B = values
    g.a_cap_words = ((3 * RF + 63) & ~(size_t)63);
    g.b_cap_words = 0;
    for (uint32_t i = 0; i < NI; ++i) { g.a_cap_words += ((size_t)g.items[i].P + 63) & ~(size_t)63; g.b_cap_words += ((size_t)g.items[i].P * g.items[i].words + 63) & ~(size_t)63; }
    if (g.a_cap_words >= (1ull << 32) || g.b_cap_words >= (1ull << 32)) return fail(DMI_ERR_INVALID_ARGUMENT, "build group too large");
    // device memory: what outlives the call (the arenas) separately from the scratch
    g.built = std::make_shared<BuiltGroup>();
    BuiltGroup& bg = *g.built;
    bg.device = device; bg.stream = g.S;
    bg.a_bytes = g.a_cap_words * 4; bg.b_off = align256(bg.a_bytes); bg.b_bytes = g.b_cap_words * 4;
    bg.keep.init(device, g.S, bg.b_off + bg.b_bytes + 4096);
    bg.d_base = bg.keep.take<uint8_t>(bg.b_off + bg.b_bytes + 256);
    const size_t parts = scan_partials_words((uint32_t)(std::max({AP, PT, RF}) + 1));
    const size_t scratch_words = (any_narrow ? 3 * RF : 0) + vtab_words + ptab_words + 5 * AP + 2 + 4 * PT + 2 + RF + 1 + 3 * RF + parts + 64 * 16;
    g.scratch.init(device, g.S, g.up_bytes + scratch_words * 4 + (size_t)M * (sizeof(MbMesh) + sizeof(MbMeshOut) + sizeof(MbWiden)) + (size_t)NI * (sizeof(MbItem) + sizeof(MbItemOut)) + ((size_t)1 << 20));
    uint8_t* d_up = g.scratch.take<uint8_t>(g.up_bytes);
    MbArgs& a = g.args;
    a.M = M; a.n_items = NI; a.total_faces = (uint32_t)RF; a.total_points = (uint32_t)PT; a.total_ap = (uint32_t)AP;
    uint32_t* d_wide = any_narrow ? g.scratch.take<uint32_t>(3 * RF) : nullptr;
    a.vtab = g.scratch.take<uint32_t>(vtab_words); a.ptab = g.scratch.take<uint32_t>(ptab_words);
    a.vslot = g.scratch.take<uint32_t>(AP); a.vflag = g.scratch.take<uint32_t>(AP + 1); a.vid = g.scratch.take<uint32_t>(AP); a.vfirst = g.scratch.take<uint32_t>(AP); a.vused = g.scratch.take<uint32_t>(AP + 1);
    a.pslot = g.scratch.take<uint32_t>(PT); a.prep = g.scratch.take<uint32_t>(PT); a.pflag = g.scratch.take<uint32_t>(PT + 1); a.used = g.scratch.take<uint32_t>(PT + 1);
    a.keep = g.scratch.take<uint32_t>(RF + 1); a.tmp_faces = g.scratch.take<uint32_t>(3 * RF); a.scan_partials = g.scratch.take<uint32_t>(parts);
    MbMesh* d_meshes = g.scratch.take<MbMesh>(M);
    MbItem* d_items = g.scratch.take<MbItem>(NI);
    a.mesh_out = g.scratch.take<MbMeshOut>(M); a.item_out = g.scratch.take<MbItemOut>(NI); a.totals = g.scratch.take<uint32_t>(4);
    MbWiden* d_widen = any_narrow ? g.scratch.take<MbWiden>(M) : nullptr;
    if (!bg.d_base || !d_up || (any_narrow && (!d_wide || !d_widen)) || !a.vtab || !a.ptab || !a.vslot || !a.vflag || !a.vid || !a.vfirst || !a.vused || !a.pslot || !a.prep || !a.pflag || !a.used ||
        !a.keep || !a.tmp_faces || !a.scan_partials || !d_meshes || !d_items || !a.mesh_out || !a.item_out || !a.totals)
      return fail(DMI_ERR_OUT_OF_MEMORY, "hipMalloc (device mesh build)");
    a.meshes = d_meshes; a.items = d_items;
    a.raw_values = reinterpret_cast<const uint32_t*>(d_up);
    a.arena_a = reinterpret_cast<uint32_t*>(bg.d_base); a.arena_b = reinterpret_cast<uint32_t*>(bg.d_base + bg.b_off);
    // pinned staging: the upload (released when the group is done) and what stays (arena A's host copy, arena B's on request, the counts)
    const size_t desc_bytes = align256((size_t)M * sizeof(MbMesh)) + align256((size_t)NI * sizeof(MbItem)) + align256((size_t)M * sizeof(MbWiden));
    const size_t stage_data = g.in_place ? 0 : g.up_bytes;   // (in place: only the descriptors go through staging)
    g.up_stage = acquire_stage(device, stage_data + desc_bytes);
    const size_t keep_counts = align256((size_t)M * sizeof(MbMeshOut)) + align256((size_t)NI * sizeof(MbItemOut)) + 256;
    const size_t keep_a = align256(bg.a_bytes), keep_b = host_values ? align256(bg.b_bytes) : 0;
    bg.stage = acquire_stage(device, keep_a + keep_b + keep_counts);
    if (!g.up_stage || !bg.stage) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (mesh build staging)");
    bg.h_a = bg.stage->p; bg.h_b = host_values ? bg.stage->p + keep_a : nullptr;
    g.h_mesh_out = reinterpret_cast<MbMeshOut*>(bg.stage->p + keep_a + keep_b);
    g.h_item_out = reinterpret_cast<MbItemOut*>(reinterpret_cast<uint8_t*>(g.h_mesh_out) + align256((size_t)M * sizeof(MbMeshOut)));
    g.h_totals = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(g.h_item_out) + align256((size_t)NI * sizeof(MbItemOut)));
    uint8_t* hp = g.up_stage->p;
    // pack: one task per accessor / index array, large ones in slices of 8 MiB
    struct Task { uint32_t k, i; size_t lo, hi; };   // i == n_items of the mesh: its indices
    std::vector<Task> tasks;
    constexpr size_t kSlice = (size_t)8 << 20;
    for (uint32_t k = 0; k < M; ++k) {
      const dmi_raw_mesh& m = raw[g.which[k]];
      for (uint32_t i = 0; i <= m.n_atts; ++i) {
        const size_t bytes = i < m.n_atts ? (size_t)m.atts[i].count * m.atts[i].num_components * 4 : (size_t)m.num_faces * 3 * component_bytes(m.index_type);
        const size_t row = i < m.n_atts ? (size_t)m.atts[i].num_components * 4 : 4;
        const size_t step = kSlice / row * row;
        for (size_t lo = 0; lo < bytes; lo += step) tasks.push_back({k, i, lo, std::min(bytes, lo + step)});
      }
    }
    const double p0 = ms();
    if (g.in_place) tasks.clear();
    if ((rc = run_parallel((uint32_t)tasks.size(), pack_threads, [&](uint32_t t) -> int {
          const Task& tk = tasks[t];
          const dmi_raw_mesh& m = raw[g.which[tk.k]];
          if (tk.i < m.n_atts) pack_rows(hp + g.row_at[g.meshes[tk.k].item0 + tk.i], m.atts[tk.i], (size_t)m.atts[tk.i].num_components * 4, tk.lo, tk.hi);
          else if (dbg_on(DMI_DBG_NO_STREAM_COPY)) std::memcpy(hp + g.idx_at[tk.k] + tk.lo, static_cast<const uint8_t*>(m.indices) + tk.lo, tk.hi - tk.lo);
          else stream_copy(hp + g.idx_at[tk.k] + tk.lo, static_cast<const uint8_t*>(m.indices) + tk.lo, tk.hi - tk.lo);
          return DMI_OK;
        }, [&](uint32_t t) { return (uint64_t)(tasks[t].hi - tasks[t].lo); }))) return rc;
    t_pack += ms() - p0;
    // descriptors ride behind the data in the same staging
    uint8_t* h_desc = hp + stage_data;
    std::memcpy(h_desc, g.meshes.data(), (size_t)M * sizeof(MbMesh));
    std::memcpy(h_desc + align256((size_t)M * sizeof(MbMesh)), g.items.data(), (size_t)NI * sizeof(MbItem));
    uint32_t n_widen = 0, widen_total = 0;
    if (any_narrow) {
      MbWiden* w = reinterpret_cast<MbWiden*>(h_desc + align256((size_t)M * sizeof(MbMesh)) + align256((size_t)NI * sizeof(MbItem)));
      for (uint32_t k = 0; k < M; ++k) {
        w[n_widen++] = MbWiden{widen_total, (uint32_t)component_bytes(raw[g.which[k]].index_type), 3u * g.meshes[k].face_off, 0u, (uint64_t)g.idx_at[k]};
        widen_total += 3u * g.meshes[k].F;
      }
    }
    if (g.in_place) {
      for (const BuildGroup::Span& sp : g.spans) { HIP_TRY(hipMemcpyAsync(d_up + sp.dev_off, reinterpret_cast<const void*>(sp.lo), sp.hi - sp.lo, hipMemcpyHostToDevice, g.S)); bytes_up += sp.hi - sp.lo; }
    } else {
      HIP_TRY(hipMemcpyAsync(d_up, hp, g.up_bytes, hipMemcpyHostToDevice, g.S));
      bytes_up += g.up_bytes;
    }
    HIP_TRY(hipMemcpyAsync(d_meshes, h_desc, (size_t)M * sizeof(MbMesh), hipMemcpyHostToDevice, g.S));
    HIP_TRY(hipMemcpyAsync(d_items, h_desc + align256((size_t)M * sizeof(MbMesh)), (size_t)NI * sizeof(MbItem), hipMemcpyHostToDevice, g.S));
    if (n_widen) HIP_TRY(hipMemcpyAsync(d_widen, h_desc + align256((size_t)M * sizeof(MbMesh)) + align256((size_t)NI * sizeof(MbItem)), (size_t)n_widen * sizeof(MbWiden), hipMemcpyHostToDevice, g.S));
    HIP_TRY(hipEventCreate(&g.ev_k0)); HIP_TRY(hipEventCreate(&g.ev_k1));
    HIP_TRY(hipEventRecord(g.ev_k0, g.S));
    if (any_narrow) {
      launch_widen_indices(d_widen, n_widen, widen_total, d_up, d_wide, g.S);
      a.raw_faces = d_wide;
    } else {
      a.raw_faces = reinterpret_cast<const uint32_t*>(d_up + g.idx_at[0]);
    }
    HIP_TRY(mesh_build_clear(a, vtab_words, ptab_words, g.S));
    launch_mesh_build(a, g.S);
    HIP_TRY(hipEventRecord(g.ev_k1, g.S));
    HIP_TRY(hipMemcpyAsync(g.h_mesh_out, a.mesh_out, (size_t)M * sizeof(MbMeshOut), hipMemcpyDeviceToHost, g.S));
    HIP_TRY(hipMemcpyAsync(g.h_item_out, a.item_out, (size_t)NI * sizeof(MbItemOut), hipMemcpyDeviceToHost, g.S));
    HIP_TRY(hipMemcpyAsync(g.h_totals, a.totals, 16, hipMemcpyDeviceToHost, g.S));
    HIP_TRY(hipEventCreateWithFlags(&g.ev_counts, long_wait_flags()));
    HIP_TRY(hipEventRecord(g.ev_counts, g.S));
    return DMI_OK;
  };
  // ---- counts are in: fetch exactly what was produced ----
  auto fetch = [&](size_t gi) -> int {
    BuildGroup& g = *groups[gi];
    BuiltGroup& bg = *g.built;
    HIP_TRY(long_wait_event(g.ev_counts));
    const size_t a_bytes = (size_t)g.h_totals[0] * 4, b_bytes = (size_t)g.h_totals[1] * 4;
    if (a_bytes > bg.a_bytes || b_bytes > bg.b_bytes) return fail(DMI_ERR_HIP, "device mesh build: arena overflow");
    if (a_bytes) HIP_TRY(hipMemcpyAsync(bg.h_a, bg.d_base, a_bytes, hipMemcpyDeviceToHost, g.S));
    if (g.host_values && b_bytes) HIP_TRY(hipMemcpyAsync(bg.h_b, bg.d_base + bg.b_off, b_bytes, hipMemcpyDeviceToHost, g.S));
    bytes_down += a_bytes + (g.host_values ? b_bytes : 0);
    HIP_TRY(hipEventCreateWithFlags(&g.ev_done, long_wait_flags()));
    HIP_TRY(hipEventRecord(g.ev_done, g.S));
    return DMI_OK;
  };
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    if ((rc = issue(gi))) return rc;
    if (gi > 0 && (rc = fetch(gi - 1))) return rc;
  }
  if (!groups.empty() && (rc = fetch(groups.size() - 1))) return rc;
  const double t_issued = ms();

  // ---- while the device works: the primitives the host builder takes ----
  if ((rc = run_parallel((uint32_t)host_list.size(), n_threads, [&](uint32_t k) -> int { return build_on_host(raw[host_list[k]], &out[host_list[k]]); },
                         [&](uint32_t k) { return (uint64_t)raw[host_list[k]].num_faces; }))) return rc;
  uint32_t n_host = (uint32_t)host_list.size(), n_device = 0, n_in_place = 0;
  double kernels_ms = 0;

  // ---- views ----
  std::vector<uint32_t> redo;   // flagged by the kernels: host builder
  for (auto& gp : groups) {
    BuildGroup& g = *gp;
    BuiltGroup& bg = *g.built;
    HIP_TRY(long_wait_event(g.ev_done));
    // (the build's scratch and staging go back without waiting for the stream again: the connectivity stage issued behind the build below would
    //  otherwise hold this thread — 14 ms per 256-mesh stage with attribute tables — until ITS read-back has landed)
    g.settled = true; g.scratch.owner_waits = true;
    float km = 0;
    if (hipEventElapsedTime(&km, g.ev_k0, g.ev_k1) == hipSuccess) kernels_ms += km; else (void)hipGetLastError();
    const uint32_t M = (uint32_t)g.meshes.size();
    bg.members.clear();
    uint64_t faces_seen = 0;
    for (uint32_t k = 0; k < M; ++k) {
      const uint32_t j = g.which[k];
      const dmi_raw_mesh& m = raw[j];
      const MbMeshOut& mo = g.h_mesh_out[k];
      BuiltGroup::Member mem;
      mem.F = mo.F_out; mem.P = mo.P_out; mem.raw_faces = m.num_faces;
      mem.faces_off = (size_t)mo.face_out_off * 12;
      if (mo.face_out_off != faces_seen) return fail(DMI_ERR_HIP, "device mesh build: face offsets out of order");
      faces_seen += mo.F_out;
      // attribute order of the built mesh: Position swapped to slot 0 (builder.rs:115-125), ids = add order
      std::vector<uint32_t> order(m.n_atts);
      for (uint32_t i = 0; i < m.n_atts; ++i) order[i] = i;
      for (uint32_t i = 0; i < m.n_atts; ++i) if (m.atts[i].att_type == DMI_ATT_POSITION) { std::swap(order[0], order[i]); break; }
      mem.atts.resize(m.n_atts);
      for (uint32_t s = 0; s < m.n_atts; ++s) {
        const MbItemOut& io = g.h_item_out[g.meshes[k].item0 + order[s]];
        mem.atts[s].val_off = bg.b_off + (size_t)io.val_off * 4;
        mem.atts[s].map_off = io.has_map ? (size_t)io.map_off * 4 : (size_t)-1;
        mem.atts[s].n_unique = io.n_out;
        mem.atts[s].att_type = m.atts[order[s]].att_type;
      }
      bg.members.push_back(std::move(mem));
      if (mo.flags) { redo.push_back(j); continue; }   // (its slot in the arena stays: the connectivity kernels skip nothing, the caller's list decides)
      std::unique_ptr<BuiltDevice> o(new BuiltDevice());
      o->group = g.built;
      o->member = k;
      o->views.resize(m.n_atts);
      o->parents.resize(m.n_atts);
      const BuiltGroup::Member& bm = bg.members.back();
      for (uint32_t s = 0; s < m.n_atts; ++s) {
        const dmi_raw_accessor& ra = m.atts[order[s]];
        dmi_attribute& v = o->views[s];
        v.values = g.host_values ? bg.h_b + (bm.atts[s].val_off - bg.b_off) : nullptr;
        v.num_unique = bm.atts[s].n_unique;
        v.component_type = ra.component_type; v.num_components = ra.num_components; v.att_type = ra.att_type; v.domain = ra.domain;
        v.unique_id = order[s];
        v.parent_index = -1;
        if (ra.num_parents) for (uint32_t q = 0; q < m.n_atts; ++q) if (order[q] == ra.parents[0]) v.parent_index = (int32_t)q;
        v.point_to_value = bm.atts[s].map_off != (size_t)-1 ? reinterpret_cast<const uint32_t*>(bg.h_a + bm.atts[s].map_off) : nullptr;
        v.num_points = bm.P;
      }
      out[j].mesh.faces = reinterpret_cast<const uint32_t*>(bg.h_a + bm.faces_off);
      out[j].mesh.num_faces = bm.F;
      out[j].mesh.atts = o->views.data();
      out[j].mesh.num_atts = m.n_atts;
      out[j].owner = static_cast<BuiltBase*>(o.release());
      ++n_device;
      if (g.in_place) ++n_in_place;
    }
    bg.total_faces = faces_seen;
    // the universal corner tables of the group's meshes, right behind the build: on the host by the time dmi_built_meshes_prepare walks them
    // (a mesh of 2^20 faces or more goes through the single-mesh prepare, which builds its own)
    bool any_large = false;
    for (const auto& mem : bg.members) any_large = any_large || mem.F >= kDeviceRelabelMinFaces;
    if (!any_large && faces_seen && !dbg_on(DMI_DBG_HOST_CONNECTIVITY) && (rc = built_group_issue_tables(bg, g.S))) return rc;
    bg.stream = nullptr; bg.keep.pool.stream = nullptr; bg.keep.owner_waits = true;   // (a null-stream synchronisation would wait for every other stage of a pipeline)   // (everything of the build has arrived; the library stream belongs to this thread, the group may outlive it)
    release_stage(g.up_stage); g.up_stage = nullptr;
  }
  groups.clear();   // (scratch back to the chunk cache; the arenas live on with their members)
  if ((rc = run_parallel((uint32_t)redo.size(), n_threads, [&](uint32_t k) -> int { return build_on_host(raw[redo[k]], &out[redo[k]]); }, nullptr))) return rc;
  n_host += (uint32_t)redo.size();
  cleanup.armed = false;
  g_last_build = dmi_build_timings{(float)t_pack, (float)kernels_ms, (float)ms(), n_device, n_host, bytes_up, bytes_down, n_in_place, 0u};
  if (trace || dbg_on(DMI_DBG_TRACE_STAGES)) std::fprintf(stderr, "[dmi] meshes_build: %u primitives (%u on the device — %u read in place —, %u on the host): pack %.2f ms, issued by %.2f, kernels %.2f, total %.2f; %.1f MB up, %.1f MB down\n",
                          n, n_device, n_in_place, n_host, t_pack, t_issued, kernels_ms, ms(), bytes_up / 1e6, bytes_down / 1e6);
  return DMI_OK;
}

}  // extern "C"
