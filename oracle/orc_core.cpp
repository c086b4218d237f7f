// oracle/orc_core.cpp — TEST INFRASTRUCTURE (see oracle.hpp header).  Byte/bit writers,
// attribute value dedup, MeshBuilder, OBJ loader.  Citations: /root/reference/draco-oxide/src/...
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <unordered_map>

#include "oracle.hpp"

namespace orc {

// utils/bit_coder.rs:20-33
void leb128_write(u64 value, Bytes& w) {
  for (;;) {
    u8 byte = (u8)(value & 0x7F);
    value >>= 7;
    if (value == 0) { w.w8(byte); break; }
    w.w8(byte | 0x80);
  }
}

// core/bit_coder.rs:113-178 with Order = LsbFirst
void BitWriterLsb::write_bits(u8 size, u64 value) {
  u8 offset = 0;
  if (pos != 0) {
    u8 rem = (u8)(8 - pos);
    if (size <= rem) {
      cur |= (u8)(value << pos);
      if (size == rem) { buf.w8(cur); cur = 0; pos = 0; } else { pos = (u8)(pos + size); }
      return;
    }
    cur |= (u8)(value << pos);
    buf.w8(cur);
    cur = 0;
    offset = rem;
  }
  int n = (size - offset) >> 3;
  for (int i = 0; i < n; ++i) { buf.w8((u8)(value >> offset)); offset = (u8)(offset + 8); }
  cur = (u8)(offset >= 64 ? 0 : (value >> offset));
  pos = (u8)((size - offset) & 7);
}
void BitWriterLsb::finish() { if (pos > 0) buf.w8(cur); pos = 0; cur = 0; }   // :181-188

// core/bit_coder.rs:113-178 with Order = MsbFirst
void BitWriterMsb::write_bits(u8 size, u64 value) {
  u8 offset = size;
  if (pos != 0) {
    u8 rem = (u8)(8 - pos);
    if (size <= rem) {
      cur |= (u8)((value & ((1ull << rem) - 1)) << (rem - size));
      if (size == rem) { buf.w8(cur); cur = 0; pos = 0; } else { pos = (u8)(pos + size); }
      return;
    }
    cur |= (u8)((value >> (size - rem)) & ((1ull << rem) - 1));
    buf.w8(cur);
    cur = 0;
    offset = (u8)(size - rem);
  }
  int n = offset >> 3;
  for (int i = 0; i < n; ++i) { offset = (u8)(offset - 8); buf.w8((u8)(value >> offset)); }
  cur = (u8)((value & ((1ull << offset) - 1)) << (8 - offset));
  pos = offset;
}
void BitWriterMsb::finish() { if (pos > 0) buf.w8(cur); pos = 0; cur = 0; }

int comp_size(CompType t) {   // core/attribute/mod.rs:534-549
  switch (t) {
    case U8: case I8: return 1;
    case U16: case I16: return 2;
    case U32: case I32: case F32: return 4;
    case U64: case I64: case F64: return 8;
  }
  return 0;
}

u32 Attribute::val_idx(u32 p) const {   // core/attribute/mod.rs:230-246
  if ((size_t)p >= len()) { std::fprintf(stderr, "oracle: point index %u out of bounds (len %zu)\n", p, len()); std::abort(); }
  return has_map ? p2v[p] : p;
}

// i32 zig-zag, utils/mod.rs:152-158 (release-mode wrapping arithmetic, quirk Q18)
i32 to_positive_i32(i32 v) {
  if (v >= 0) return (i32)((u32)v << 1);
  return (i32)((((u32)(-(v + 1))) << 1) + 1u);
}

// ---------------------------------------------------------------------------------------------
// Value equality used by remove_duplicate_values: NdVector PartialEq = component-wise `==`
// (macros/draco-nd-vector/src/lib.rs:167-175).  For floats: -0.0 == 0.0, NaN != NaN (quirk Q19).
// ---------------------------------------------------------------------------------------------
static bool value_eq(const Attribute& a, size_t i, size_t j) {
  const u8* pi = a.data.data() + i * a.value_size();
  const u8* pj = a.data.data() + j * a.value_size();
  for (int c = 0; c < a.ncomp; ++c) {
    switch (a.ctype) {
      case F32: { float x, y; std::memcpy(&x, pi + 4 * c, 4); std::memcpy(&y, pj + 4 * c, 4); if (!(x == y)) return false; break; }
      case F64: { double x, y; std::memcpy(&x, pi + 8 * c, 8); std::memcpy(&y, pj + 8 * c, 8); if (!(x == y)) return false; break; }
      default: { int s = comp_size(a.ctype); if (std::memcmp(pi + s * c, pj + s * c, s) != 0) return false; }
    }
  }
  return true;
}

static void erase_values(Attribute& a, const std::vector<size_t>& sorted_dups) {
  // self.buffer.remove(i) back-to-front (:448-451) == keep everything not in the list, in order.
  const size_t vs = a.value_size();
  const size_t n = a.num_unique();
  std::vector<u8> out;
  out.reserve(a.data.size());
  size_t k = 0;
  for (size_t i = 0; i < n; ++i) {
    if (k < sorted_dups.size() && sorted_dups[k] == i) { ++k; continue; }
    out.insert(out.end(), a.data.begin() + i * vs, a.data.begin() + (i + 1) * vs);
  }
  a.data.swap(out);
}

// core/attribute/mod.rs:394-452.  faithful=true restates the two quadratic loops literally;
// faithful=false computes the same result (first-occurrence numbering) with a hash map.
void remove_duplicate_values(Attribute& a, bool faithful) {
  const size_t n = a.len();   // at this point there is no map, so len() == num_unique()
  std::vector<u32> map(n);
  for (size_t i = 0; i < n; ++i) map[i] = (u32)i;
  std::vector<size_t> dups;
  if (faithful) {
    std::vector<u8> is_dup(n, 0);
    for (size_t i = 0; i < n; ++i) {
      if (i == n - 1) break;                       // :401-404
      if (is_dup[i]) continue;                     // :405-408 (duplicate_indeces.contains(&i))
      std::vector<size_t> local;
      for (size_t j = i + 1; j < n; ++j) if (value_eq(a, i, j)) local.push_back(j);   // :410-414
      if (local.empty()) continue;
      for (size_t d : local) { map[d] = (u32)i; is_dup[d] = 1; }
      dups.insert(dups.end(), local.begin(), local.end());
    }
    // :426-443 literal renumbering
    size_t curr_max = 0;
    for (size_t p = 0; p < n; ++p) {
      size_t val_idx = map[p];
      if (val_idx == curr_max + 1) {
        curr_max += 1;
      } else if (val_idx > curr_max + 1) {
        curr_max += 1;
        for (size_t q = p; q < n; ++q) if (map[q] == val_idx) map[q] = (u32)curr_max;
      }
    }
  } else {
    // key = canonical bytes (-0.0 → +0.0); values containing NaN never compare equal.
    const size_t vs = a.value_size();
    std::unordered_map<std::string, u32> first;
    first.reserve(n * 2);
    u32 next_rank = 0;
    std::string key(vs, '\0');
    for (size_t i = 0; i < n; ++i) {
      const u8* p = a.data.data() + i * vs;
      bool has_nan = false;
      std::memcpy(&key[0], p, vs);
      if (a.ctype == F32) {
        for (int c = 0; c < a.ncomp; ++c) { float x; std::memcpy(&x, p + 4 * c, 4); if (x != x) has_nan = true; if (x == 0.0f) { x = 0.0f; std::memcpy(&key[4 * c], &x, 4); } }
      } else if (a.ctype == F64) {
        for (int c = 0; c < a.ncomp; ++c) { double x; std::memcpy(&x, p + 8 * c, 8); if (x != x) has_nan = true; if (x == 0.0) { x = 0.0; std::memcpy(&key[8 * c], &x, 8); } }
      }
      if (has_nan) { map[i] = next_rank++; continue; }
      auto it = first.find(key);
      if (it == first.end()) { first.emplace(key, next_rank); map[i] = next_rank++; }
      else { map[i] = it->second; dups.push_back(i); }
    }
  }
  if (!dups.empty()) { a.has_map = true; a.p2v = map; }   // :444-446
  std::sort(dups.begin(), dups.end());                      // :448
  erase_values(a, dups);
}

// Attribute::remove (core/attribute/mod.rs:454-483) applied to a set of points, expressed as
// its net effect: drop the points from the map; unique values that lose their last referencing
// point are removed and higher indices shifted down (order of the remaining values preserved).
static void remove_points(Attribute& a, const std::vector<u8>& drop /*len = a.len()*/) {
  const size_t n = a.len();
  if (a.has_map) {
    std::vector<u32> kept_map;
    kept_map.reserve(n);
    std::vector<u8> used(a.num_unique(), 0);
    for (size_t p = 0; p < n; ++p) if (!drop[p]) { kept_map.push_back(a.p2v[p]); used[a.p2v[p]] = 1; }
    std::vector<u32> renum(a.num_unique(), 0);
    std::vector<size_t> gone;
    u32 k = 0;
    for (size_t v = 0; v < used.size(); ++v) { if (used[v]) renum[v] = k++; else gone.push_back(v); }
    for (auto& v : kept_map) v = renum[v];
    erase_values(a, gone);
    a.p2v.swap(kept_map);
  } else {
    std::vector<size_t> gone;
    for (size_t p = 0; p < n; ++p) if (drop[p]) gone.push_back(p);
    erase_values(a, gone);   // remove_unique_val (:499-507)
  }
}

// core/mesh/builder.rs:62-90
std::string mesh_build(std::vector<Attribute> atts, std::vector<std::array<u32, 3>> faces, bool faithful, Mesh& out) {
  (void)faithful;
  // dependency_check :95-111 — TextureCoordinate must list a Position parent (core/attribute/mod.rs:624-637)
  for (auto& a : atts) {
    if (a.type == TextureCoordinate) {
      bool ok = false;
      for (u32 pid : a.parents) for (auto& b : atts) if (b.id == pid && b.type == Position) ok = true;
      if (!ok) return "MinimumDependencyError(TextureCoordinate, Position)";
    }
  }
  // get_sorted_attributes :115-125 — swap the Position attribute to slot 0 (ids unchanged)
  for (size_t i = 0; i < atts.size(); ++i) if (atts[i].type == Position) { std::swap(atts[0], atts[i]); break; }

  // deduplicate_vertices_based_on_positions :194-250
  if (!atts.empty()) {
    u32 maxp = 0;
    for (auto& f : faces) for (u32 p : f) maxp = std::max(maxp, p);
    const size_t num_vertices = (size_t)maxp + 1;
    // hash_vertex :254-279, as it stands: for every attribute with point < len() the key takes (attribute type, component type, N)
    // and the RAW BYTES of the point's unique value (`get_data_as_bytes()[value_idx * size ..]`), not the value index.  After value
    // dedup the two agree for every row that compares equal to itself (`==` rows share the first occurrence's bytes; rows that are
    // `!=` differ in some byte) — but a row holding a NaN equals nothing in remove_duplicate_values (mod.rs:394-452) and so keeps a
    // unique value of its own, while two byte-identical NaN rows still hash equal here: such points MERGE.  (The reference keeps a
    // 64-bit SipHash of the key; a collision of that hash is the one thing not restated.)
    std::unordered_map<std::string, u32> uniq;
    std::vector<u32> mapping(num_vertices);
    u32 unique_count = 0;
    std::string key;
    for (size_t p = 0; p < num_vertices; ++p) {
      key.clear();
      for (auto& a : atts) {
        if (p >= a.len()) { key.push_back('\0'); continue; }   // :258 — the attribute takes no part
        key.push_back('\1');
        key.push_back((char)a.type); key.push_back((char)a.ctype); key.push_back((char)a.ncomp);
        const size_t vs = a.value_size();
        key.append(reinterpret_cast<const char*>(a.data.data() + (size_t)a.val_idx((u32)p) * vs), vs);
      }
      auto it = uniq.find(key);
      if (it != uniq.end()) mapping[p] = it->second;
      else { uniq.emplace(key, unique_count); mapping[p] = unique_count++; }
    }
    if (unique_count != num_vertices) {
      // remap_attribute :283-371: drop every point that is not the first occurrence of its class
      for (auto& a : atts) {
        if (unique_count == a.len()) continue;   // :285-287
        std::vector<u8> met(unique_count, 0), drop(a.len(), 0);
        for (size_t v = 0; v < mapping.size(); ++v) {
          bool rm;
          if (met[mapping[v]]) rm = true; else { met[mapping[v]] = 1; rm = false; }
          if (rm && v < a.len()) drop[v] = 1;
        }
        remove_points(a, drop);
      }
      for (auto& f : faces) for (u32& p : f) p = mapping[p];
    }
  }
  // :77-79 remove degenerate faces
  {
    std::vector<std::array<u32, 3>> kept;
    kept.reserve(faces.size());
    for (auto& f : faces) if (f[0] != f[1] && f[1] != f[2] && f[2] != f[0]) kept.push_back(f);
    faces.swap(kept);
  }
  // remove_unused_vertices :129-189
  if (!faces.empty() && !atts.empty()) {
    u32 maxp = 0;
    for (auto& f : faces) for (u32 p : f) maxp = std::max(maxp, p);
    std::vector<u8> used((size_t)maxp + 1, 0);
    for (auto& f : faces) for (u32 p : f) used[p] = 1;
    bool any_unused = false;
    for (u8 u : used) if (!u) any_unused = true;
    for (auto& a : atts) {
      if (a.len() <= used.size() && !any_unused) continue;
      std::vector<u8> drop(a.len(), 0);
      for (size_t p = 0; p < a.len(); ++p) drop[p] = (p >= used.size()) ? 1 : !used[p];
      remove_points(a, drop);
    }
    std::vector<u32> offsets(used.size());
    u32 removed = 0;
    for (size_t v = 0; v < used.size(); ++v) { offsets[v] = removed; if (!used[v]) ++removed; }
    for (auto& f : faces) for (u32& p : f) p -= offsets[p];
  }
  out.atts = std::move(atts);
  out.faces = std::move(faces);
  return "";
}

// ---------------------------------------------------------------------------------------------
// OBJ loader: io/obj/mod.rs:14-42 on top of tobj 4.0.3 (Cargo.toml:24-25; not vendored) with
// LoadOptions{triangulate:true, single_index:true}.  tobj behaviour restated from its published
// algorithm: one running (v,vt,vn)→index map per model in first-appearance order; polygons
// are fan-triangulated (a, b_i, b_{i+1}); only models[0] is used by the reference.
// Pinned by io/obj/mod.rs:73-88 (tetrahedron) only.
// ---------------------------------------------------------------------------------------------
std::string load_obj(const std::string& path, bool faithful, Mesh& out) {
  std::ifstream in(path);
  if (!in) return "cannot open " + path;
  std::vector<float> v, vt, vn;
  struct Key { long a, b, c; bool operator<(const Key& o) const { return a != o.a ? a < o.a : b != o.b ? b < o.b : c < o.c; } };
  std::map<Key, u32> index_map;
  std::vector<float> pos, tex, nor;
  std::vector<u32> indices;
  bool model_closed = false;
  std::string line;
  auto add_vertex = [&](const Key& k) {
    auto it = index_map.find(k);
    if (it != index_map.end()) { indices.push_back(it->second); return; }
    u32 next = (u32)index_map.size();
    pos.push_back(v[3 * k.a]); pos.push_back(v[3 * k.a + 1]); pos.push_back(v[3 * k.a + 2]);
    if (!vt.empty() && k.b >= 0) { tex.push_back(vt[2 * k.b]); tex.push_back(vt[2 * k.b + 1]); }
    if (!vn.empty() && k.c >= 0) { nor.push_back(vn[3 * k.c]); nor.push_back(vn[3 * k.c + 1]); nor.push_back(vn[3 * k.c + 2]); }
    indices.push_back(next);
    index_map.emplace(k, next);
  };
  while (std::getline(in, line)) {
    size_t h = line.find('#');
    if (h != std::string::npos) line.resize(h);
    std::istringstream ss(line);
    std::string tag;
    if (!(ss >> tag)) continue;
    if (tag == "v") { std::string t; for (int i = 0; i < 3; ++i) { ss >> t; v.push_back(std::strtof(t.c_str(), nullptr)); } }
    else if (tag == "vt") { std::string t; for (int i = 0; i < 2; ++i) { if (ss >> t) vt.push_back(std::strtof(t.c_str(), nullptr)); else vt.push_back(0.0f); } }
    else if (tag == "vn") { std::string t; for (int i = 0; i < 3; ++i) { ss >> t; vn.push_back(std::strtof(t.c_str(), nullptr)); } }
    else if (tag == "f") {
      if (model_closed) continue;
      std::vector<Key> face;
      std::string t;
      while (ss >> t) {
        Key k{-1, -1, -1};
        long vals[3] = {0, 0, 0};
        bool have[3] = {false, false, false};
        int fi = 0;
        size_t s = 0;
        for (size_t i = 0; i <= t.size() && fi < 3; ++i) {
          if (i == t.size() || t[i] == '/') {
            if (i > s) { vals[fi] = std::strtol(t.substr(s, i - s).c_str(), nullptr, 10); have[fi] = true; }
            ++fi; s = i + 1;
          }
        }
        auto fix = [](long idx, size_t n) -> long { return idx > 0 ? idx - 1 : (long)n + idx; };
        if (have[0]) k.a = fix(vals[0], v.size() / 3);
        if (have[1]) k.b = fix(vals[1], vt.size() / 2);
        if (have[2]) k.c = fix(vals[2], vn.size() / 3);
        face.push_back(k);
      }
      if (face.size() < 3) continue;
      Key a = face[0];
      Key b = face[1];
      for (size_t i = 2; i < face.size(); ++i) { add_vertex(a); add_vertex(b); add_vertex(face[i]); b = face[i]; }
    } else if (tag == "o" || tag == "g" || tag == "usemtl") {
      // tobj starts a new model here if the current one already has faces; the reference reads models[0] only.
      if (!indices.empty()) model_closed = true;
    }
  }
  // io/obj/mod.rs:23-40
  std::vector<Attribute> atts;
  auto make = [&](u32 id, AttType ty, Domain d, int n, const std::vector<float>& vals, std::vector<u32> parents) {
    Attribute a;
    a.id = id; a.type = ty; a.domain = d; a.ctype = F32; a.ncomp = n; a.parents = std::move(parents);
    a.data.resize(vals.size() * 4);
    if (!vals.empty()) std::memcpy(a.data.data(), vals.data(), a.data.size());
    remove_duplicate_values(a, faithful);   // Attribute::from, core/attribute/mod.rs:87-103
    atts.push_back(std::move(a));
  };
  u32 id = 0;
  make(id++, Position, DomPosition, 3, pos, {});
  if (!nor.empty()) make(id++, Normal, DomCorner, 3, nor, {0});
  if (!tex.empty()) make(id++, TextureCoordinate, DomCorner, 2, tex, {0});
  std::vector<std::array<u32, 3>> faces(indices.size() / 3);
  for (size_t f = 0; f < faces.size(); ++f) faces[f] = {indices[3 * f], indices[3 * f + 1], indices[3 * f + 2]};
  return mesh_build(std::move(atts), std::move(faces), faithful, out);
}

}  // namespace orc
