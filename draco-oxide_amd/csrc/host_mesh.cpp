// host_mesh.cpp — MeshBuilder::build (core/mesh/builder.rs:62-90) and Attribute::from's value dedup
// (core/attribute/mod.rs:394-452) on flat arrays, behind dmi_mesh_build.  Host only, no GPU.
//
// The reference's builders are O(V²) (pairwise value compare, `contains` on the duplicate list, Vec::remove per
// point); this produces the identical mesh — same unique-value order (first occurrence), same point_to_value maps,
// same surviving points and faces — with open-addressing class tables (a slot holds a ROW INDEX; rows are inserted
// in index order, so the row that owns a slot is the first occurrence of its class; the slots of a block of rows are
// prefetched before the block is inserted) and one compaction pass per step:
//   1. per attribute (one host thread each): value dedup with `==` semantics (f32/f64: -0.0 == 0.0, a row holding a
//      NaN equals nothing); the map exists only when a duplicate was found (mod.rs:444-446)
//   2. the Position attribute is swapped to slot 0 (builder.rs:115-125; ids keep the add order)
//   3. points that agree in every attribute are merged.  builder.rs:254-279 hashes (type, component type, N) and the
//      RAW BYTES of each attribute's unique value at the point.  For a row that equals itself that is the value
//      index (`==` rows share the first occurrence's bytes, `!=` rows differ in a byte); a row holding a NaN has a
//      unique value of its own, yet two byte-identical NaN rows hash equal — their points merge when every other
//      attribute agrees too.  The merge key of an attribute at a point is therefore its BYTE CLASS: the value index,
//      except that a NaN row takes the index of the first byte-identical NaN row (`cls`).  The first point of each
//      class survives (remap_attribute :283-371 removes the others through Attribute::remove, whose net effect is:
//      the point leaves the map, a value that loses its last point leaves the buffer, order preserved)
//   4. degenerate faces are dropped (:77-79)
//   5. points no face references are removed the same way and the faces renumbered (:129-189)
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/draco_mi.h"
#include "dmi_host.hpp"

namespace dmi {
namespace {

struct BuiltAtt {
  std::vector<uint8_t> values;   // num_unique rows
  size_t value_size = 0;
  std::vector<uint32_t> p2v;     // valid when has_map
  bool has_map = false;
  uint32_t len = 0;              // number of points
  uint8_t component_type = 0, num_components = 0, att_type = 0, domain = 0;
  uint32_t id = 0;
  std::vector<uint32_t> parents;
  uint32_t num_unique() const { return value_size ? (uint32_t)(values.size() / value_size) : 0u; }
  uint32_t val_idx(uint32_t p) const { return has_map ? p2v[p] : p; }
};

size_t component_size(uint8_t t) {
  switch (t) {
    case DMI_U8: case DMI_I8: return 1;
    case DMI_U16: case DMI_I16: return 2;
    case DMI_U32: case DMI_I32: case DMI_F32: return 4;
    case DMI_U64: case DMI_I64: case DMI_F64: return 8;
    default: return 0;
  }
}

// keep the rows whose flag is set, in order
void keep_values(BuiltAtt& a, const std::vector<uint8_t>& keep) {
  const size_t vs = a.value_size;
  size_t w = 0;
  for (size_t v = 0; v < keep.size(); ++v) {
    if (!keep[v]) continue;
    if (w != v) std::memmove(a.values.data() + w * vs, a.values.data() + v * vs, vs);
    ++w;
  }
  a.values.resize(w * vs);
}

inline uint64_t mix64(uint64_t h, uint64_t w) {
  h ^= w;
  h *= 0xFF51AFD7ED558CCDull;
  h ^= h >> 32;
  h *= 0xC4CEB9FE1A85EC53ull;
  return h ^ (h >> 29);
}
inline size_t table_slots(size_t n) { size_t s = 16; while (s < 2 * n) s <<= 1; return s; }   // at most half full

// `==` of two rows neither of which holds a NaN; hash of the canonical row (a zero of either sign hashes as +0)
struct RowOps {
  const uint8_t* base;
  size_t vs;
  int kind;   // 0 = bytes, 4 = f32 words, 8 = f64 words
  unsigned n;
  const uint8_t* row(size_t i) const { return base + i * vs; }
  bool has_nan(const uint8_t* r) const {
    if (kind == 4) { for (unsigned c = 0; c < n; ++c) { uint32_t w; std::memcpy(&w, r + 4 * c, 4); if ((w & 0x7FFFFFFFu) > 0x7F800000u) return true; } }
    else if (kind == 8) { for (unsigned c = 0; c < n; ++c) { uint64_t w; std::memcpy(&w, r + 8 * c, 8); if ((w & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) return true; } }
    return false;
  }
  uint64_t hash(const uint8_t* r) const {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    if (kind == 4) { for (unsigned c = 0; c < n; ++c) { uint32_t w; std::memcpy(&w, r + 4 * c, 4); if ((w << 1) == 0u) w = 0u; h = mix64(h, w); } }
    else if (kind == 8) { for (unsigned c = 0; c < n; ++c) { uint64_t w; std::memcpy(&w, r + 8 * c, 8); if ((w << 1) == 0ull) w = 0ull; h = mix64(h, w); } }
    else { size_t k = 0; for (; k + 8 <= vs; k += 8) { uint64_t w; std::memcpy(&w, r + k, 8); h = mix64(h, w); } if (k < vs) { uint64_t w = 0; std::memcpy(&w, r + k, vs - k); h = mix64(h, w ^ ((uint64_t)(vs - k) << 56)); } }
    return h;
  }
  bool equal(const uint8_t* a, const uint8_t* b) const {
    if (kind == 4) { for (unsigned c = 0; c < n; ++c) { uint32_t x, y; std::memcpy(&x, a + 4 * c, 4); std::memcpy(&y, b + 4 * c, 4); if (x != y && ((x | y) << 1) != 0u) return false; } return true; }
    if (kind == 8) { for (unsigned c = 0; c < n; ++c) { uint64_t x, y; std::memcpy(&x, a + 8 * c, 8); std::memcpy(&y, b + 8 * c, 8); if (x != y && ((x | y) << 1) != 0ull) return false; } return true; }
    return std::memcmp(a, b, vs) == 0;
  }
};

// core/attribute/mod.rs:394-452.  cls: see the file comment (left empty when no two NaN rows are byte-identical).
void dedup_values(BuiltAtt& a, std::vector<uint32_t>& cls) {
  const size_t n = a.num_unique(), vs = a.value_size;
  if (n == 0) return;
  const RowOps ops{a.values.data(), vs, a.component_type == DMI_F32 ? 4 : a.component_type == DMI_F64 ? 8 : 0, a.num_components};
  std::vector<uint32_t> map = VecPool<uint32_t>::get().take(n);   // (becomes the attribute's point → value map when a duplicate is found)
  map.resize(n);
  std::vector<uint8_t> keep(n, 1);
  const size_t slots = table_slots(n), mask = slots - 1;
  Pooled<uint32_t> table(slots, kNone);
  uint32_t* tab = table.v.data();
  std::unordered_map<std::string, uint32_t> nan_first;   // bytes of a NaN row → value index of the first such row
  std::vector<std::pair<uint32_t, uint32_t>> nan_again;  // (row, value index of the first byte-identical NaN row)
  constexpr size_t kAhead = 32;
  uint64_t hs[kAhead];
  uint32_t next = 0;
  bool any_dup = false;
  for (size_t lo = 0; lo < n; lo += kAhead) {
    const size_t hi = std::min(n, lo + kAhead);
    for (size_t i = lo; i < hi; ++i) { hs[i - lo] = ops.hash(ops.row(i)) & mask; __builtin_prefetch(tab + hs[i - lo], 1); }
    for (size_t i = lo; i < hi; ++i) {
      const uint8_t* row = ops.row(i);
      if (ops.has_nan(row)) {   // NaN != NaN: never a duplicate, never a representative
        auto ins = nan_first.emplace(std::string(reinterpret_cast<const char*>(row), vs), next);
        if (!ins.second) nan_again.emplace_back((uint32_t)i, ins.first->second);
        map[i] = next++;
        continue;
      }
      size_t h = hs[i - lo];
      for (;; h = (h + 1) & mask) {
        const uint32_t s = tab[h];
        if (s == kNone) { tab[h] = (uint32_t)i; map[i] = next++; break; }
        if (ops.equal(ops.row(s), row)) { map[i] = map[s]; keep[i] = 0; any_dup = true; break; }
      }
    }
  }
  if (!nan_again.empty()) { cls = map; for (auto& e : nan_again) cls[e.first] = e.second; }
  if (!any_dup) { pool_give(map); return; }
  a.has_map = true;
  a.p2v.swap(map);
  keep_values(a, keep);
}

// Net effect of Attribute::remove (mod.rs:454-483) over a set of points
void remove_points(BuiltAtt& a, const std::vector<uint8_t>& drop) {
  if (a.has_map) {
    std::vector<uint32_t> kept;
    kept.reserve(a.len);
    std::vector<uint8_t> used(a.num_unique(), 0);
    for (uint32_t p = 0; p < a.len; ++p) if (!drop[p]) { kept.push_back(a.p2v[p]); used[a.p2v[p]] = 1; }
    std::vector<uint32_t> renum(used.size(), 0);
    uint32_t k = 0;
    for (size_t v = 0; v < used.size(); ++v) if (used[v]) renum[v] = k++;
    for (auto& v : kept) v = renum[v];
    keep_values(a, used);
    a.p2v.swap(kept);
    a.len = (uint32_t)a.p2v.size();
  } else {
    std::vector<uint8_t> keep(a.len);
    uint32_t k = 0;
    for (uint32_t p = 0; p < a.len; ++p) { keep[p] = !drop[p]; k += keep[p]; }
    keep_values(a, keep);
    a.len = k;
  }
}

struct BuiltOwner : BuiltBase {
  std::vector<BuiltAtt> atts;
  std::vector<uint32_t> faces;
  std::vector<dmi_attribute> views;
  ~BuiltOwner() override {   // the large arrays go back to the pool: a fresh 100 MB vector costs more in page faults than the pass that fills it
    pool_give(faces);
    for (auto& a : atts) { pool_give(a.values); pool_give(a.p2v); }
  }
};

// fn(k) for every attribute, side by side when the arrays are large
template <class Fn>
int over_attributes(size_t n_atts, size_t rows, Fn&& fn) {
  if (n_atts > 1 && rows >= (1u << 16) && host_threads() > 1) return guarded_pool(n_atts, host_threads(), fn);
  for (size_t k = 0; k < n_atts; ++k) fn(k);
  return 0;
}

}  // namespace
}  // namespace dmi

using namespace dmi;

extern "C" {

int dmi_mesh_build(const dmi_raw_attribute* in, uint32_t n_atts, const uint32_t* faces_in, uint32_t num_faces, dmi_built_mesh* out) {
  if (!out || (!in && n_atts) || (!faces_in && num_faces)) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  std::unique_ptr<BuiltOwner> o(new BuiltOwner());
  const bool trace = dbg_on(DMI_DBG_BUILD_TRACE);
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { if (!trace) return; const auto t1 = std::chrono::steady_clock::now(); std::fprintf(stderr, "[dmi_mesh_build] %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; };
  o->atts.resize(n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) {
    BuiltAtt& a = o->atts[i];
    const dmi_raw_attribute& r = in[i];
    const size_t cs = component_size(r.component_type);
    if (!cs || r.num_components == 0) return host_fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "attribute " + std::to_string(i) + ": bad component type / count");
    if (!r.data && r.num_points) return host_fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + ": no data");
    a.value_size = cs * r.num_components;
    pool_fit(a.values, (size_t)r.num_points * a.value_size);
    a.values.resize((size_t)r.num_points * a.value_size);
    if (r.num_points) { const uint8_t* src = static_cast<const uint8_t*>(r.data); uint8_t* dst = a.values.data(); parallel_for(a.values.size(), [&](size_t lo, size_t hi) { std::memcpy(dst + lo, src + lo, hi - lo); }); }
    a.len = r.num_points;
    a.component_type = r.component_type; a.num_components = r.num_components; a.att_type = r.att_type; a.domain = r.domain;
    a.id = i;
    for (uint32_t k = 0; k < r.num_parents; ++k) {
      if (r.parents[k] >= n_atts) return host_fail(DMI_ERR_BAD_PARENT, "attribute " + std::to_string(i) + ": parent id out of range");
      a.parents.push_back(r.parents[k]);
    }
  }
  // dependency_check (builder.rs:95-111): a TextureCoordinate attribute needs a Position parent (mod.rs:624-637)
  for (auto& a : o->atts) {
    if (a.att_type != DMI_ATT_TEXCOORD) continue;
    bool ok = false;
    for (uint32_t pid : a.parents) if (in[pid].att_type == DMI_ATT_POSITION) ok = true;
    if (!ok) return host_fail(DMI_ERR_BAD_PARENT, "MinimumDependencyError(TextureCoordinate, Position)");
  }
  lap("copy");
  std::vector<std::vector<uint32_t>> cls(n_atts);   // byte classes where they differ from the value indices (NaN rows)
  {
    size_t rows = 0;
    for (auto& a : o->atts) rows += a.len;
    const int st = over_attributes(n_atts, rows, [&](size_t i) { dedup_values(o->atts[i], cls[i]); });
    if (st) return host_fail(DMI_ERR_INVALID_ARGUMENT, st == 1 ? "out of memory" : "mesh build failed");
  }
  for (size_t i = 0; i < o->atts.size(); ++i) if (o->atts[i].att_type == DMI_ATT_POSITION) { std::swap(o->atts[0], o->atts[i]); std::swap(cls[0], cls[i]); break; }

  lap("values");
  std::vector<uint32_t>& faces = o->faces;
  pool_fit(faces, (size_t)num_faces * 3);
  faces.resize((size_t)num_faces * 3);
  uint32_t maxp_in = 0;
  {
    std::mutex fold;
    parallel_for(faces.size(), [&](size_t lo, size_t hi) {
      uint32_t m = 0;
      for (size_t k = lo; k < hi; ++k) { const uint32_t p = faces_in[k]; faces[k] = p; m = std::max(m, p); }
      std::lock_guard<std::mutex> lock(fold);
      maxp_in = std::max(maxp_in, m);
    });
  }
  // deduplicate_vertices_based_on_positions (:194-250)
  if (!o->atts.empty()) {
    const uint32_t maxp = maxp_in;
    const size_t num_vertices = (size_t)maxp + 1;   // (1 for an empty face list, like the reference's unwrap_or(0) + 1)
    const size_t K = o->atts.size();
    // an attribute in which every point below num_vertices carries a value of its own keeps all points apart: nothing merges
    bool all_apart = false;
    for (size_t k = 0; k < K; ++k) if (!o->atts[k].has_map && cls[k].empty() && o->atts[k].len >= num_vertices) all_apart = true;
    if (!all_apart) {
      // the key of point p: per attribute its byte class, kNone where the attribute is shorter than p (it takes no part, :258; the
      // constant (type, component type, N) triple adds nothing to a comparison between points of one mesh)
      Pooled<uint32_t> keys_store(num_vertices * K, 0u);
      uint32_t* keys = keys_store.v.data();
      for (size_t k = 0; k < K; ++k) {
        const BuiltAtt& a = o->atts[k];
        const uint32_t* c = !cls[k].empty() ? cls[k].data() : a.has_map ? a.p2v.data() : nullptr;
        const size_t m = std::min<size_t>(num_vertices, a.len);
        for (size_t p = 0; p < m; ++p) keys[p * K + k] = c ? c[p] : (uint32_t)p;
        for (size_t p = m; p < num_vertices; ++p) keys[p * K + k] = kNone;
      }
      const size_t slots = table_slots(num_vertices), mask = slots - 1;
      Pooled<uint32_t> table(slots, kNone);
      uint32_t* tab = table.v.data();
      std::vector<uint32_t> mapping(num_vertices);
      uint32_t unique_count = 0;
      constexpr size_t kAhead = 32;
      uint64_t hs[kAhead];
      for (size_t lo = 0; lo < num_vertices; lo += kAhead) {
        const size_t hi = std::min(num_vertices, lo + kAhead);
        for (size_t p = lo; p < hi; ++p) {
          uint64_t h = 0x9E3779B97F4A7C15ull;
          for (size_t k = 0; k < K; ++k) h = mix64(h, keys[p * K + k]);
          hs[p - lo] = h & mask;
          __builtin_prefetch(tab + hs[p - lo], 1);
        }
        for (size_t p = lo; p < hi; ++p) {
          for (size_t h = hs[p - lo];; h = (h + 1) & mask) {
            const uint32_t s = tab[h];
            if (s == kNone) { tab[h] = (uint32_t)p; mapping[p] = unique_count++; break; }
            if (std::memcmp(keys + (size_t)s * K, keys + p * K, 4 * K) == 0) { mapping[p] = mapping[s]; break; }
          }
        }
      }
      if (unique_count != num_vertices) {
        // (mapping[v] counts the classes in first-occurrence order: v is a later member of its class ⇔ its class id is below the running count)
        std::vector<uint8_t> later(num_vertices);
        { uint32_t seen = 0; for (size_t v = 0; v < num_vertices; ++v) { later[v] = mapping[v] < seen; seen += !later[v]; } }
        over_attributes(o->atts.size(), num_vertices, [&](size_t k) {
          BuiltAtt& a = o->atts[k];
          if (unique_count == a.len) return;   // :285-287
          std::vector<uint8_t> drop(a.len, 0);
          std::memcpy(drop.data(), later.data(), std::min<size_t>(a.len, num_vertices));
          remove_points(a, drop);
        });
        parallel_for(faces.size(), [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) faces[k] = mapping[faces[k]]; });
      }
    }
  }
  lap("points");
  uint32_t maxp_kept = 0;
  {   // degenerate faces (:77-79)
    size_t w = 0;
    const size_t n = faces.size();
    uint32_t* f = faces.data();
    for (size_t r = 0; r + 2 < n; r += 3) {
      const uint32_t a = f[r], b = f[r + 1], c = f[r + 2];
      if (a != b && b != c && c != a) {
        if (w != r) { f[w] = a; f[w + 1] = b; f[w + 2] = c; }
        w += 3;
        maxp_kept = std::max(maxp_kept, std::max(a, std::max(b, c)));
      }
    }
    faces.resize(w);
  }
  // remove_unused_vertices (:129-189)
  if (!faces.empty() && !o->atts.empty()) {
    const uint32_t maxp = maxp_kept;
    std::vector<uint8_t> used((size_t)maxp + 1, 0);
    for (uint32_t p : faces) used[p] = 1;
    size_t n_unused = 0;
    for (uint8_t u : used) n_unused += !u;
    const bool any_unused = n_unused != 0;
    over_attributes(o->atts.size(), used.size(), [&](size_t k) {
      BuiltAtt& a = o->atts[k];
      if (a.len <= used.size() && !any_unused) return;
      std::vector<uint8_t> drop(a.len, 0);
      for (uint32_t p = 0; p < a.len; ++p) drop[p] = (p >= used.size()) ? 1 : !used[p];
      remove_points(a, drop);
    });
    if (any_unused) {
      std::vector<uint32_t> offsets(used.size());
      uint32_t removed = 0;
      for (size_t v = 0; v < used.size(); ++v) { offsets[v] = removed; if (!used[v]) ++removed; }
      parallel_for(faces.size(), [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) faces[k] -= offsets[faces[k]]; });
    }
  }
  lap("faces");
  // views
  o->views.resize(o->atts.size());
  for (size_t i = 0; i < o->atts.size(); ++i) {
    const BuiltAtt& a = o->atts[i];
    dmi_attribute& v = o->views[i];
    v.values = a.values.data();
    v.num_unique = a.num_unique();
    v.component_type = a.component_type; v.num_components = a.num_components; v.att_type = a.att_type; v.domain = a.domain;
    v.unique_id = a.id;
    v.parent_index = -1;
    if (!a.parents.empty()) for (size_t k = 0; k < o->atts.size(); ++k) if (o->atts[k].id == a.parents[0]) v.parent_index = (int32_t)k;
    v.point_to_value = a.has_map ? a.p2v.data() : nullptr;
    v.num_points = a.len;
  }
  out->mesh.faces = faces.data();
  out->mesh.num_faces = (uint32_t)(faces.size() / 3);
  out->mesh.atts = o->views.data();
  out->mesh.num_atts = (uint32_t)o->views.size();
  out->owner = static_cast<BuiltBase*>(o.release());
  return DMI_OK;
}

void dmi_built_mesh_free(dmi_built_mesh* m) {
  if (!m || !m->owner) return;
  delete static_cast<BuiltBase*>(m->owner);
  m->owner = nullptr;
  m->mesh = dmi_mesh{};
}

}  // extern "C"
