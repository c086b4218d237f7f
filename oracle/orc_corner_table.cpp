// oracle/orc_corner_table.cpp — TEST INFRASTRUCTURE (see oracle.hpp header).
// Literal restatement of core/corner_table/mod.rs and attribute_corner_table.rs.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "oracle.hpp"

namespace orc {

// core/corner_table/mod.rs:84-118
std::string CornerTable::build(const std::vector<std::array<u32, 3>>& faces, const Attribute& pos) {
  mesh_faces = &faces;
  conn_faces.resize(faces.size());
  for (size_t f = 0; f < faces.size(); ++f)
    for (int k = 0; k < 3; ++k) conn_faces[f][k] = pos.val_idx(faces[f][k]);   // :85-93
  ncorners = (u32)faces.size() * 3;
  nverts = 0;
  opposite_corners.clear();
  left_most_corners.clear();
  c2v_override.assign(ncorners, NONE);
  non_manifold_vertex_parents.clear();

  // get_unused_vertices :236-250 → panic at :105-108
  {
    u32 maxv = 0;
    for (auto& f : conn_faces) for (u32 v : f) maxv = std::max(maxv, v);
    std::vector<u8> used((size_t)maxv + 1, 0);
    for (auto& f : conn_faces) for (u32 v : f) used[v] = 1;
    for (u8 u : used) if (!u) return "Mesh contains unused vertices";
  }
  compute_table();
  if (contains_non_manifold_edges()) handle_non_manifold_edges();
  compute_left_most_corners();
  return "";
}

// :121-145
bool CornerTable::contains_non_manifold_edges() {
  std::vector<std::array<u32, 2>> edges;
  edges.reserve(conn_faces.size() * 3);
  for (auto& f : conn_faces) {
    edges.push_back({f[0], f[1]}); edges.push_back({f[1], f[2]}); edges.push_back({f[2], f[0]});
  }
  for (auto& e : edges) if (e[0] > e[1]) std::swap(e[0], e[1]);
  std::sort(edges.begin(), edges.end());
  int count = 1;
  for (size_t i = 1; i < edges.size(); ++i) {
    if (edges[i] == edges[i - 1]) { if (++count > 2) return true; } else count = 1;
  }
  return false;
}

// :149-234
void CornerTable::handle_non_manifold_edges() {
  std::vector<u8> visited(ncorners, 0);
  std::vector<std::pair<u32, u32>> sinks;
  for (;;) {
    bool connectivity_updated = false;
    for (u32 c0 = 0; c0 < ncorners; ++c0) {
      if (visited[c0]) continue;
      u32 c = c0;
      sinks.clear();
      u32 first_c = c, curr_c = c;
      for (;;) {
        u32 nx = swing_left(curr_c);
        if (nx == NONE) break;
        if (nx == first_c || visited[nx]) break;
        curr_c = nx;
      }
      first_c = curr_c;
      for (;;) {
        visited[curr_c] = 1;
        u32 sink_c = next(curr_c);
        u32 sink_v = vertex_idx(sink_c);
        u32 edge_c = previous(curr_c);
        bool vertex_connectivity_updated = false;
        for (auto& as : sinks) {
          if (as.first == sink_v) {
            u32 other_edge_c = as.second;
            u32 opp_edge_c = opposite(edge_c);
            if (opp_edge_c != NONE && opp_edge_c == other_edge_c) continue;
            u32 opp_other_edge_c = opposite(other_edge_c);
            if (opp_edge_c != NONE) opposite_corners[opp_edge_c] = NONE;
            if (opp_other_edge_c != NONE) opposite_corners[opp_other_edge_c] = NONE;
            opposite_corners[edge_c] = NONE;
            opposite_corners[other_edge_c] = NONE;
            vertex_connectivity_updated = true;
            break;
          }
        }
        if (vertex_connectivity_updated) { connectivity_updated = true; break; }
        sinks.emplace_back(vertex_idx(previous(curr_c)), sink_c);
        u32 r = swing_right(curr_c);
        if (r == NONE) break;
        curr_c = r;
        if (curr_c == first_c) break;
      }
    }
    if (!connectivity_updated) break;
  }
}

// :252-340 — Draco's half-edge bucket matching, including the non-advancing `continue` (quirk Q22).
void CornerTable::compute_table() {
  opposite_corners.assign(ncorners, NONE);
  std::vector<u32> num_corners_on_vertices;
  num_corners_on_vertices.reserve(ncorners);
  for (u32 c = 0; c < ncorners; ++c) {
    u32 v1 = vertex_idx(c);
    if (v1 >= num_corners_on_vertices.size()) num_corners_on_vertices.resize((size_t)v1 + 1, 0);
    num_corners_on_vertices[v1] += 1;
  }
  std::vector<std::pair<u32, u32>> vertex_edges(ncorners, {NONE, NONE});   // (sink vertex, edge corner)
  std::vector<u32> vertex_offset(num_corners_on_vertices.size());
  {
    u32 off = 0;
    for (size_t i = 0; i < num_corners_on_vertices.size(); ++i) { vertex_offset[i] = off; off += num_corners_on_vertices[i]; }
  }
  for (u32 c = 0; c < ncorners; ++c) {
    u32 tip_v = vertex_idx(c);
    u32 source_v = vertex_idx(next(c));
    u32 sink_v = vertex_idx(previous(c));
    u32 f_idx = c / 3;
    if (c == f_idx * 3) {   // :289-295 only the first corner of a degenerate face is skipped
      u32 v0 = vertex_idx(c);
      if (v0 == source_v || v0 == sink_v || source_v == sink_v) continue;
    }
    u32 opposite_c = NONE;
    u32 n_on_vert = num_corners_on_vertices[sink_v];
    u32 offset = vertex_offset[sink_v];
    for (u32 i = 0; i < n_on_vert; ++i) {
      u32 other_v = vertex_edges[offset].first;
      if (other_v == NONE) break;
      if (other_v == source_v) {
        if (tip_v == vertex_idx(vertex_edges[offset].second)) continue;   // :308-310 (offset NOT advanced)
        opposite_c = vertex_edges[offset].second;
        for (u32 j = i + 1; j < n_on_vert; ++j) {
          vertex_edges[offset] = vertex_edges[offset + 1];
          if (vertex_edges[offset].first == NONE) break;
          offset += 1;
        }
        vertex_edges[offset].first = NONE;
        break;
      }
      offset += 1;
    }
    if (opposite_c == NONE) {
      u32 n_src = num_corners_on_vertices[source_v];
      u32 first_c = vertex_offset[source_v];
      for (u32 corner = first_c; corner < n_src + first_c; ++corner) {
        if (vertex_edges[corner].first == NONE) {
          vertex_edges[corner].first = sink_v;
          vertex_edges[corner].second = c;
          break;
        }
      }
    } else {
      opposite_corners[c] = opposite_c;
      opposite_corners[opposite_c] = c;
    }
  }
  nverts = (u32)num_corners_on_vertices.size();
}

// :342-416
void CornerTable::compute_left_most_corners() {
  left_most_corners.assign(nverts, NONE);
  std::vector<u8> visited_vertices(nverts, 0);
  std::vector<u8> visited_corners(ncorners, 0);
  const u32 nfaces = num_faces();
  for (u32 f = 0; f < nfaces; ++f) {
    for (u32 i = 0; i < 3; ++i) {
      u32 c = 3 * f + i;
      if (visited_corners[c]) continue;
      u32 v = vertex_idx(c);
      bool is_non_manifold_vertex = false;
      if (visited_vertices[v]) {
        left_most_corners.push_back(NONE);
        non_manifold_vertex_parents.push_back(v);
        visited_vertices.push_back(0);
        v = nverts;
        nverts += 1;
        is_non_manifold_vertex = true;
      }
      visited_vertices[v] = 1;
      visited_corners[c] = 1;
      left_most_corners[v] = c;
      if (is_non_manifold_vertex) c2v_override[c] = v;
      u32 act_c = swing_left(c);
      while (act_c != NONE) {
        if (act_c == c) break;
        visited_corners[act_c] = 1;
        left_most_corners[v] = act_c;
        if (is_non_manifold_vertex) c2v_override[act_c] = v;
        act_c = swing_left(act_c);
      }
      if (act_c == NONE) {
        act_c = c;
        while (act_c != NONE) {
          visited_corners[act_c] = 1;
          if (is_non_manifold_vertex) c2v_override[act_c] = v;
          act_c = swing_right(act_c);
        }
      }
    }
  }
}

// attribute_corner_table.rs:16-77
void AttributeCornerTable::build(const CornerTable& ct, const Attribute& att) {
  is_edge_on_seam.assign(ct.num_corners(), 0);
  is_vertex_on_seam.assign(ct.num_vertices(), 0);
  for (u32 c = 0; c < ct.num_corners(); ++c) {
    u32 opp = ct.opposite(c);
    if (opp == NONE) {
      is_edge_on_seam[c] = 1;
      is_vertex_on_seam[ct.vertex_idx(CornerTable::next(c))] = 1;
      is_vertex_on_seam[ct.vertex_idx(CornerTable::previous(c))] = 1;
      continue;
    }
    if (opp < c) continue;
    u32 c1 = c, c2 = opp;
    for (int k = 0; k < 2; ++k) {
      c1 = CornerTable::next(c1);
      c2 = CornerTable::previous(c2);
      u32 i1 = ct.point_idx(c1);
      u32 i2 = ct.point_idx(c2);
      if (att.val_idx(i1) != att.val_idx(i2)) {
        is_edge_on_seam[c] = 1;
        is_edge_on_seam[opp] = 1;
        is_vertex_on_seam[ct.vertex_idx(CornerTable::next(c))] = 1;
        is_vertex_on_seam[ct.vertex_idx(CornerTable::previous(c))] = 1;
        is_vertex_on_seam[ct.vertex_idx(CornerTable::next(opp))] = 1;
        is_vertex_on_seam[ct.vertex_idx(CornerTable::previous(opp))] = 1;
        break;
      }
    }
  }
  corner_to_vertex.assign(ct.num_corners(), 0);
  nverts = ct.num_vertices();
  recompute_vertices(att, ct);
}

// attribute_corner_table.rs:79-137
void AttributeCornerTable::recompute_vertices(const Attribute& att, const CornerTable& ct) {
  vertex_to_attribute_map.clear();
  left_most_corners.clear();
  u32 num_new_vertices = 0;
  for (u32 v = 0; v < ct.num_vertices(); ++v) {
    u32 c = ct.left_most_corner(v);
    u32 first_vert_id = num_new_vertices++;
    vertex_to_attribute_map.push_back(att.val_idx(ct.point_idx(c)));
    u32 first_c = c;
    if (is_vertex_on_seam[v]) {
      u32 curr = swing_left(first_c, ct);
      while (curr != NONE) {
        first_c = curr;
        if (curr == c) { std::fprintf(stderr, "oracle: unreachable (attribute_corner_table.rs:108)\n"); std::abort(); }
        curr = swing_left(curr, ct);
      }
    }
    corner_to_vertex[first_c] = first_vert_id;
    left_most_corners.push_back(first_c);
    u32 curr = ct.swing_right(first_c);
    while (curr != NONE) {
      if (curr == first_c) break;
      if (is_edge_on_seam[CornerTable::next(curr)]) {
        first_vert_id = num_new_vertices++;
        vertex_to_attribute_map.push_back(att.val_idx(ct.point_idx(curr)));
        left_most_corners.push_back(curr);
      }
      corner_to_vertex[curr] = first_vert_id;
      curr = ct.swing_right(curr);
    }
  }
  nverts = num_new_vertices;
}

}  // namespace orc
