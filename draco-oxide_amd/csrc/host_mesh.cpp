// host_mesh.cpp — MeshBuilder::build (core/mesh/builder.rs:62-90) and Attribute::from's value dedup
// (core/attribute/mod.rs:394-452) on flat arrays, behind dmi_mesh_build.  Host only, no GPU.
//
// The reference's builders are O(V²) (pairwise value compare, `contains` on the duplicate list, Vec::remove per
// point); this produces the identical mesh — same unique-value order (first occurrence), same point_to_value maps,
// same surviving points and faces — with hash maps and one compaction pass per step:
//   1. per attribute: value dedup with `==` semantics (f32/f64: -0.0 == 0.0, a row holding a NaN equals nothing);
//      the map exists only when a duplicate was found (mod.rs:444-446)
//   2. the Position attribute is swapped to slot 0 (builder.rs:115-125; ids keep the add order)
//   3. points that agree in every attribute are merged: builder.rs:254-279 hashes the raw bytes of each attribute's
//      UNIQUE value at the point, which after step 1 is the same as comparing the tuple of value indices; the first
//      point of each class survives (remap_attribute :283-371 removes the others through Attribute::remove, whose net
//      effect is: the point leaves the map, a value that loses its last point leaves the buffer, order preserved)
//   4. degenerate faces are dropped (:77-79)
//   5. points no face references are removed the same way and the faces renumbered (:129-189)
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
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

// core/attribute/mod.rs:394-452
void dedup_values(BuiltAtt& a) {
  const size_t n = a.num_unique(), vs = a.value_size;
  if (n == 0) return;
  std::vector<uint32_t> map(n);
  std::vector<uint8_t> keep(n, 1);
  std::unordered_map<std::string, uint32_t> first;
  first.reserve(n * 2);
  std::string key(vs, '\0');
  uint32_t next = 0;
  bool any_dup = false;
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* row = a.values.data() + i * vs;
    std::memcpy(&key[0], row, vs);
    bool has_nan = false;
    if (a.component_type == DMI_F32) {
      for (int c = 0; c < a.num_components; ++c) {
        float x; std::memcpy(&x, row + 4 * c, 4);
        if (x != x) has_nan = true;
        if (x == 0.0f) { x = 0.0f; std::memcpy(&key[4 * c], &x, 4); }   // -0.0 == 0.0
      }
    } else if (a.component_type == DMI_F64) {
      for (int c = 0; c < a.num_components; ++c) {
        double x; std::memcpy(&x, row + 8 * c, 8);
        if (x != x) has_nan = true;
        if (x == 0.0) { x = 0.0; std::memcpy(&key[8 * c], &x, 8); }
      }
    }
    if (has_nan) { map[i] = next++; continue; }   // NaN != NaN: never a duplicate, never a representative
    auto it = first.find(key);
    if (it == first.end()) { first.emplace(key, next); map[i] = next++; }
    else { map[i] = it->second; keep[i] = 0; any_dup = true; }
  }
  if (!any_dup) return;
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
};

}  // namespace
}  // namespace dmi

using namespace dmi;

extern "C" {

int dmi_mesh_build(const dmi_raw_attribute* in, uint32_t n_atts, const uint32_t* faces_in, uint32_t num_faces, dmi_built_mesh* out) {
  if (!out || (!in && n_atts) || (!faces_in && num_faces)) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  std::unique_ptr<BuiltOwner> o(new BuiltOwner());
  o->atts.resize(n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) {
    BuiltAtt& a = o->atts[i];
    const dmi_raw_attribute& r = in[i];
    const size_t cs = component_size(r.component_type);
    if (!cs || r.num_components == 0) return host_fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "attribute " + std::to_string(i) + ": bad component type / count");
    if (!r.data && r.num_points) return host_fail(DMI_ERR_INVALID_ARGUMENT, "attribute " + std::to_string(i) + ": no data");
    a.value_size = cs * r.num_components;
    a.values.assign(static_cast<const uint8_t*>(r.data), static_cast<const uint8_t*>(r.data) + (size_t)r.num_points * a.value_size);
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
  for (auto& a : o->atts) dedup_values(a);
  for (size_t i = 0; i < o->atts.size(); ++i) if (o->atts[i].att_type == DMI_ATT_POSITION) { std::swap(o->atts[0], o->atts[i]); break; }

  std::vector<uint32_t>& faces = o->faces;
  faces.assign(faces_in, faces_in + (size_t)num_faces * 3);
  // deduplicate_vertices_based_on_positions (:194-250)
  if (!o->atts.empty()) {
    uint32_t maxp = 0;
    for (uint32_t p : faces) maxp = std::max(maxp, p);
    const size_t num_vertices = (size_t)maxp + 1;   // (1 for an empty face list, like the reference's unwrap_or(0) + 1)
    std::unordered_map<std::string, uint32_t> uniq;
    uniq.reserve(num_vertices * 2);
    std::vector<uint32_t> mapping(num_vertices);
    uint32_t unique_count = 0;
    std::string key;
    for (size_t p = 0; p < num_vertices; ++p) {
      key.clear();
      for (auto& a : o->atts) {
        const uint32_t v = p < a.len ? a.val_idx((uint32_t)p) : kNone;   // attributes shorter than p do not take part (:258)
        key.append(reinterpret_cast<const char*>(&v), 4);
      }
      auto it = uniq.find(key);
      if (it != uniq.end()) mapping[p] = it->second;
      else { uniq.emplace(key, unique_count); mapping[p] = unique_count++; }
    }
    if (unique_count != num_vertices) {
      for (auto& a : o->atts) {
        if (unique_count == a.len) continue;   // :285-287
        std::vector<uint8_t> met(unique_count, 0), drop(a.len, 0);
        for (size_t v = 0; v < mapping.size(); ++v) {
          const bool again = met[mapping[v]];
          met[mapping[v]] = 1;
          if (again && v < a.len) drop[v] = 1;
        }
        remove_points(a, drop);
      }
      for (uint32_t& p : faces) p = mapping[p];
    }
  }
  {   // degenerate faces (:77-79)
    size_t w = 0;
    for (size_t f = 0; f + 2 < faces.size(); f += 3) {
      const uint32_t a = faces[f], b = faces[f + 1], c = faces[f + 2];
      if (a != b && b != c && c != a) { faces[w] = a; faces[w + 1] = b; faces[w + 2] = c; w += 3; }
    }
    faces.resize(w);
  }
  // remove_unused_vertices (:129-189)
  if (!faces.empty() && !o->atts.empty()) {
    uint32_t maxp = 0;
    for (uint32_t p : faces) maxp = std::max(maxp, p);
    std::vector<uint8_t> used((size_t)maxp + 1, 0);
    for (uint32_t p : faces) used[p] = 1;
    bool any_unused = false;
    for (uint8_t u : used) if (!u) any_unused = true;
    for (auto& a : o->atts) {
      if (a.len <= used.size() && !any_unused) continue;
      std::vector<uint8_t> drop(a.len, 0);
      for (uint32_t p = 0; p < a.len; ++p) drop[p] = (p >= used.size()) ? 1 : !used[p];
      remove_points(a, drop);
    }
    std::vector<uint32_t> offsets(used.size());
    uint32_t removed = 0;
    for (size_t v = 0; v < used.size(); ++v) { offsets[v] = removed; if (!used[v]) ++removed; }
    for (uint32_t& p : faces) p -= offsets[p];
  }
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
