// oracle/orc_edgebreaker.cpp — TEST INFRASTRUCTURE (see oracle.hpp header).
// Literal restatement of encode/connectivity/edgebreaker.rs (Standard traversal) and
// shared/attribute/sequence.rs.
#include <algorithm>
#include <map>

#include "oracle.hpp"

namespace orc {

namespace {

enum Sym : u8 { C = 0, S = 1, L = 2, R = 3, E = 4 };   // symbol_encoder.rs:5-12

struct TopologySplit { u64 merging, split; bool right; };   // shared/connectivity/edgebreaker/mod.rs:6-17

struct EB {
  const CornerTable& ct;
  std::vector<u8> visited_vertices, visited_faces, visited_holes;
  std::vector<u32> vertex_hole_id;   // NONE = not on a hole
  std::vector<u32> stack;
  u64 last_encoded_symbol_idx = ~0ull;
  std::vector<u32> processed_connectivity_corners;
  std::map<u32, u64> face_to_split_symbol_map;
  u64 num_split_symbols = 0;
  std::vector<u32> init_face_connectivity_corners;
  std::vector<TopologySplit> topology_splits;
  std::vector<u8> symbols;
  std::vector<u8> interior_cfg;
  std::string err;

  explicit EB(const CornerTable& t) : ct(t) {
    visited_vertices.assign(ct.num_vertices(), 0);
    visited_faces.assign(ct.num_faces(), 0);
  }

  // edgebreaker.rs:195-224.  NOTE: the inner loop at :213-215 is
  //     while self.corner_table.opposite(c).is_some() { c = self.corner_table.next(c); }
  // i.e. it rotates inside the same face and never crosses to the opposite face (Google Draco
  // crosses).  Consequence: every boundary vertex gets its own "hole" id.  Restated literally.
  void compute_boundaries_literal() {
    vertex_hole_id.assign(ct.num_vertices(), NONE);
    for (u32 c0 = 0; c0 < ct.num_corners(); ++c0) {
      if (ct.opposite(c0) != NONE) continue;
      u32 v = ct.vertex_idx(CornerTable::next(c0));
      if (vertex_hole_id[v] != NONE) continue;
      u32 boundary_idx = (u32)visited_holes.size();
      visited_holes.push_back(0);
      u32 c = c0;
      while (vertex_hole_id[v] == NONE) {
        vertex_hole_id[v] = boundary_idx;
        c = CornerTable::next(c);
        while (ct.opposite(c) != NONE) c = CornerTable::next(c);
        v = ct.vertex_idx(CornerTable::next(c));
      }
    }
  }

  // :226-256
  u32 process_boundary(u32 start_corner, bool encode_first_vertex) {
    u32 corner = CornerTable::previous(start_corner);
    while (ct.opposite(corner) != NONE) corner = CornerTable::next(ct.opposite(corner));
    u32 start_v = ct.vertex_idx(start_corner);
    u32 n = 0;
    if (encode_first_vertex) { visited_vertices[start_v] = 1; n += 1; }
    if (vertex_hole_id[start_v] == NONE) { err = "process_boundary: start vertex is not on a hole (reference unwrap() panic, edgebreaker.rs:243)"; return n; }
    visited_holes[vertex_hole_id[start_v]] = 1;
    u32 curr_v = ct.vertex_idx(CornerTable::previous(corner));
    while (curr_v != start_v) {
      visited_vertices[curr_v] = 1;
      n += 1;
      corner = CornerTable::next(corner);
      while (ct.opposite(corner) != NONE) corner = CornerTable::next(ct.opposite(corner));
      curr_v = ct.vertex_idx(CornerTable::previous(corner));
    }
    return n;
  }

  bool is_right_face_visited(u32 c) const { u32 r = ct.get_right_corner(c); return r == NONE ? true : visited_faces[r / 3]; }   // :355-361
  bool is_left_face_visited(u32 c) const { u32 l = ct.get_left_corner(c); return l == NONE ? true : visited_faces[l / 3]; }      // :366-372

  // :434-448
  void check_and_store_topology_split_event(u64 merging_symbol_idx, bool right, u32 split_face) {
    auto it = face_to_split_symbol_map.find(split_face);
    if (it == face_to_split_symbol_map.end()) return;
    topology_splits.push_back({merging_symbol_idx, it->second, right});
  }

  // :261-350
  void edgebreaker_from(u32 c) {
    stack.clear();
    stack.push_back(c);
    const u32 num_faces = ct.num_faces();
    while (!stack.empty()) {
      c = stack.back();
      if (visited_faces[c / 3]) { stack.pop_back(); continue; }
      u32 num_visited_faces = 0;
      while (num_visited_faces < num_faces) {
        num_visited_faces += 1;
        last_encoded_symbol_idx += 1;   // wrapping_add from usize::MAX
        u32 face_idx = c / 3;
        visited_faces[face_idx] = 1;
        processed_connectivity_corners.push_back(c);
        u32 v = ct.vertex_idx(c);
        if (!visited_vertices[v]) {
          visited_vertices[v] = 1;
          if (vertex_hole_id[v] == NONE) {
            symbols.push_back(C);
            c = ct.get_right_corner(c);
            continue;
          }
        }
        u32 right_c = ct.get_right_corner(c);
        u32 left_c = ct.get_left_corner(c);
        if (is_right_face_visited(c)) {
          if (right_c != NONE) check_and_store_topology_split_event(last_encoded_symbol_idx, true, right_c / 3);
          if (is_left_face_visited(c)) {
            if (left_c != NONE) check_and_store_topology_split_event(last_encoded_symbol_idx, false, left_c / 3);
            symbols.push_back(E);
            stack.pop_back();
            break;
          } else {
            symbols.push_back(R);
            c = left_c;
          }
        } else {
          if (is_left_face_visited(c)) {
            if (left_c != NONE) check_and_store_topology_split_event(last_encoded_symbol_idx, false, left_c / 3);
            symbols.push_back(L);
            c = right_c;
          } else {
            symbols.push_back(S);
            num_split_symbols += 1;
            u32 hole = vertex_hole_id[v];
            if (hole != NONE && !visited_holes[hole]) process_boundary(c, false);
            face_to_split_symbol_map[face_idx] = last_encoded_symbol_idx;
            stack.back() = left_c;
            stack.push_back(right_c);
            break;
          }
        }
      }
    }
  }

  // :411-431
  std::pair<bool, u32> begin_from(u32 face_idx) {
    u32 corner_index = 3 * face_idx;
    for (int k = 0; k < 3; ++k) {
      if (ct.opposite(corner_index) == NONE) return {false, corner_index};
      if (vertex_hole_id[ct.vertex_idx(corner_index)] != NONE) {
        u32 right = corner_index;
        while (right != NONE) { corner_index = right; right = ct.swing_right(right); }
        return {false, CornerTable::previous(corner_index)};
      }
      corner_index = CornerTable::next(corner_index);
    }
    return {true, corner_index};
  }
};

// zero_prob idiom used three times in edgebreaker.rs (:602, :641) — f32 arithmetic, quirk Q18 casts.
u8 zero_prob_of(size_t count0, size_t len) {
  float p = ((float)count0 / (float)len) * 256.0f + 0.5f;
  // `as u16`: saturating, NaN → 0
  u16 q;
  if (p != p) q = 0; else if (p <= 0.0f) q = 0; else if (p >= 65535.0f) q = 65535; else q = (u16)p;
  if (q < 1) q = 1;
  if (q > 255) q = 255;
  return (u8)q;
}

}  // namespace

// encode/connectivity/mod.rs:17-36 → edgebreaker.rs:128-193 (new) + :458-530 (encode_connectivity)
std::string encode_connectivity(const Mesh& mesh, Bytes& w, ConnOutput& out) {
  const Attribute* pos = nullptr;
  for (auto& a : mesh.atts) if (a.type == Position) { pos = &a; break; }
  if (!pos) return "no position attribute";
  std::string err;
  {
    StageTimer t_ct(3);
    err = out.ct.build(mesh.faces, *pos);
    if (!err.empty()) return err;
    // init_attribute_data :172-193 — one table per non-Position attribute, in attribute order
    out.att_tables.clear();
    for (auto& a : mesh.atts) {
      if (a.type == Position) continue;
      out.att_tables.emplace_back();
      out.att_tables.back().build(out.ct, a);
    }
  }
  const CornerTable& ct = out.ct;
  EB eb(ct);

  w.w8(0);   // EdgebreakerKind::Standard, :467
  eb.compute_boundaries_literal();
  leb128_write(ct.num_vertices(), w);
  leb128_write(mesh.faces.size(), w);
  w.w8((u8)out.att_tables.size());

  for (u32 c = 0; c < ct.num_corners(); ++c) {   // :478-511
    u32 face_idx = c / 3;
    if (eb.visited_faces[face_idx]) continue;
    auto [interior, start_corner] = eb.begin_from(face_idx);
    eb.interior_cfg.push_back(interior ? 1 : 0);
    if (interior) {
      u32 ci = start_corner;
      eb.visited_vertices[ct.vertex_idx(ci)] = 1;
      eb.visited_vertices[ct.vertex_idx(CornerTable::next(ci))] = 1;
      eb.visited_vertices[ct.vertex_idx(CornerTable::previous(ci))] = 1;
      eb.visited_faces[face_idx] = 1;
      eb.init_face_connectivity_corners.push_back(CornerTable::next(ci));
      u32 corner_opp = ct.opposite(CornerTable::next(ci));
      eb.edgebreaker_from(corner_opp);
    } else {
      eb.process_boundary(CornerTable::next(start_corner), true);
      eb.edgebreaker_from(start_corner);
    }
  }
  if (!eb.err.empty()) return eb.err;
  leb128_write(eb.symbols.size(), w);       // :514
  leb128_write(eb.num_split_symbols, w);    // :517

  // encode_topology_splits :375-403
  {
    u64 last_idx = 0;
    leb128_write(eb.topology_splits.size(), w);
    for (auto& s : eb.topology_splits) {
      leb128_write(s.merging - last_idx, w);
      leb128_write(s.merging - s.split, w);
      last_idx = s.merging;
    }
    BitWriterLsb bw(w);
    for (auto& s : eb.topology_splits) bw.write_bits(1, s.right ? 1 : 0);
    bw.finish();
  }

  // DefaultTraversal::encode :575-656
  {
    Bytes sym;
    {
      BitWriterLsb bw(sym);
      for (size_t i = eb.symbols.size(); i-- > 0;) {
        switch (eb.symbols[i]) {   // CrLight::encode_symbol, symbol_encoder.rs:51-58
          case C: bw.write_bits(1, 0); break;
          case S: bw.write_bits(3, 0b1); break;
          case L: bw.write_bits(3, 0b11); break;
          case R: bw.write_bits(3, 0b101); break;
          case E: bw.write_bits(3, 0b111); break;
        }
      }
      bw.finish();
    }
    leb128_write(sym.size(), w);
    w.append(sym);

    size_t c0 = 0;
    for (u8 b : eb.interior_cfg) if (!b) ++c0;
    u8 zp = zero_prob_of(c0, eb.interior_cfg.size());
    w.w8(zp);
    {
      RabsCoder rc(zp);
      for (size_t i = eb.interior_cfg.size(); i-- > 0;) rc.write(eb.interior_cfg[i] ? 1 : 0);
      Bytes b;
      std::string e = rc.flush(b);
      if (!e.empty()) return e;
      leb128_write(b.size(), w);
      w.append(b);
    }
    // attribute seams :611-653
    std::vector<u8> vf(ct.num_faces(), 0);
    std::vector<std::vector<u8>> seams(out.att_tables.size());
    for (size_t i = eb.processed_connectivity_corners.size(); i-- > 0;) {
      u32 c = eb.processed_connectivity_corners[i];
      u32 corners[3] = {c, CornerTable::next(c), CornerTable::previous(c)};
      vf[c / 3] = 1;
      for (int k = 0; k < 3; ++k) {
        u32 opp = ct.opposite(corners[k]);
        if (opp == NONE) continue;
        if (vf[opp / 3]) continue;
        for (size_t j = 0; j < out.att_tables.size(); ++j)
          seams[j].push_back(out.att_tables[j].opposite(corners[k], ct) == NONE ? 1 : 0);
      }
    }
    for (auto& sd : seams) {
      size_t z = 0;
      for (u8 s : sd) if (!s) ++z;
      u8 pz = zero_prob_of(z, sd.size());
      w.w8(pz);
      RabsCoder rc(pz);
      for (size_t i = sd.size(); i-- > 0;) rc.write(sd[i]);
      Bytes b;
      std::string e = rc.flush(b);
      if (!e.empty()) return e;
      leb128_write(b.size(), w);
      w.append(b);
    }
  }

  // :523-529
  std::reverse(eb.init_face_connectivity_corners.begin(), eb.init_face_connectivity_corners.end());
  out.corners_of_edgebreaker = eb.init_face_connectivity_corners;
  out.corners_of_edgebreaker.insert(out.corners_of_edgebreaker.end(), eb.processed_connectivity_corners.begin(), eb.processed_connectivity_corners.end());
  static const char names[] = "CSLRE";
  out.symbols.clear();
  for (u8 s : eb.symbols) out.symbols.push_back(names[s]);
  return "";
}

// shared/attribute/sequence.rs:48-151.  The `remove every stack entry lying in the current face`
// loops (:98-131) are O(stack) per face in the reference; because the face has just been marked
// visited, any such entry would be skipped when popped (:54-56), so omitting the removal
// (faithful=false) yields the identical sequence.  tests/test_oracle_kat.py checks both modes agree.
std::vector<u32> compute_sequence(const TableView& tv, std::vector<u32> stack, bool faithful) {
  std::vector<u8> visited_vertices(tv.num_vertices(), 0), visited_faces(tv.num_faces(), 0);
  std::vector<u32> out;
  out.reserve(tv.num_vertices());
  auto visit = [&](u32 v, u32 c) { if (!visited_vertices[v]) out.push_back(c); visited_vertices[v] = 1; };   // :41-46
  auto remove_face_entries = [&](u32 face_idx) {
    if (!faithful) return;
    for (size_t i = stack.size(); i-- > 0;) if (stack[i] / 3 == face_idx) stack.erase(stack.begin() + i);
  };
  while (!stack.empty()) {
    u32 curr = stack.back();
    stack.pop_back();
    u32 v = tv.vertex_idx(curr);
    if (visited_faces[curr / 3]) continue;
    u32 next_c = TableView::next(curr), prev_c = TableView::previous(curr);
    u32 next_v = tv.vertex_idx(next_c), prev_v = tv.vertex_idx(prev_c);
    if (!visited_vertices[next_v] || !visited_vertices[prev_v]) {   // :61-68
      visit(next_v, next_c);
      visit(prev_v, prev_c);
      stack.push_back(curr);
      continue;
    }
    u32 face_idx = curr / 3;
    visited_faces[face_idx] = 1;
    if (!visited_vertices[v]) {   // :76-84
      visit(v, curr);
      if (!tv.is_on_boundary(v)) { stack.push_back(tv.get_right_corner(curr)); continue; }
    }
    visit(v, curr);
    u32 right_corner = tv.get_right_corner(curr), left_corner = tv.get_left_corner(curr);
    bool right_visited = right_corner != NONE && visited_faces[right_corner / 3];
    bool left_visited = left_corner != NONE && visited_faces[left_corner / 3];
    if (right_visited) {
      if (left_visited) {
        remove_face_entries(face_idx);
      } else {
        remove_face_entries(face_idx);
        if (left_corner != NONE) stack.push_back(left_corner);
      }
    } else {
      if (left_visited) {
        remove_face_entries(face_idx);
        if (right_corner != NONE) stack.push_back(right_corner);
      } else {
        if (left_corner != NONE) stack.push_back(left_corner);
        if (right_corner != NONE) stack.push_back(right_corner);
      }
    }
  }
  return out;
}

}  // namespace orc
