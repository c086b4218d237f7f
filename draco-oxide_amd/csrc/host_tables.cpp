// host_tables.cpp — the reference's corner tables on the host: the universal table (core/corner_table/mod.rs: half-edge matching, non-manifold edges and
// vertices, left-most corners) and the attribute tables (attribute_corner_table.rs).  Flat arrays, parallel slices for large meshes.  Split out of host_conn.cpp
// in round 5; bit-identical to the reference's tables (quirks kept: SURVEY.md §8a-Q Q22 half-edge matching).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <unordered_map>
#include <sys/mman.h>

#include "dmi_host.hpp"


namespace dmi {

namespace {

struct UniversalBuilder {
  CornerTables& t;
  uint32_t C;
  explicit UniversalBuilder(CornerTables& tt) : t(tt), C(tt.F * 3) {}

  uint32_t swing_left(uint32_t c) const { uint32_t o = t.opp_own[corner_next(c)]; return o == kNone ? kNone : corner_next(o); }
  uint32_t swing_right(uint32_t c) const { uint32_t o = t.opp_own[corner_prev(c)]; return o == kNone ? kNone : corner_prev(o); }

  // Half-edge matching (core/corner_table/mod.rs:252-340).  Each vertex owns a bucket of pending
  // half-edges (sink vertex, corner) sized by its corner count; a corner looks for the reverse edge
  // in its sink's bucket.  Quirk Q22: a candidate with the same tip vertex aborts the search.
  void match_half_edges() {
    std::vector<uint32_t> count;
    count.reserve(t.V ? t.V : 16);
    for (uint32_t c = 0; c < C; ++c) {
      uint32_t v = t.c2v_own[c];
      if (v >= count.size()) count.resize((size_t)v + 1, 0);
      ++count[v];
    }
    const uint32_t nv = (uint32_t)count.size();
    std::vector<uint32_t> start(nv + 1, 0);
    for (uint32_t v = 0; v < nv; ++v) start[v + 1] = start[v] + count[v];
    std::vector<uint32_t> he_sink(C, kNone), he_corner(C, kNone);
    t.opp_own.assign(C, kNone);
    for (uint32_t c = 0; c < C; ++c) {
      const uint32_t tip = t.c2v_own[c], src = t.c2v_own[corner_next(c)], snk = t.c2v_own[corner_prev(c)];
      if (c % 3 == 0 && (tip == src || tip == snk || src == snk)) continue;   // :289-295
      uint32_t found = kNone;
      const uint32_t lo = start[snk], hi = start[snk + 1];
      for (uint32_t s = lo; s < hi; ++s) {
        if (he_sink[s] == kNone) break;
        if (he_sink[s] != src) continue;
        if (t.c2v_own[he_corner[s]] == tip) break;   // Q22: mirrored face → stop searching
        found = he_corner[s];
        uint32_t k = s;                           // delete slot s, keep order
        while (k + 1 < hi && he_sink[k + 1] != kNone) { he_sink[k] = he_sink[k + 1]; he_corner[k] = he_corner[k + 1]; ++k; }
        he_sink[k] = kNone;
        break;
      }
      if (found == kNone) {
        for (uint32_t s = start[src]; s < start[src + 1]; ++s)
          if (he_sink[s] == kNone) { he_sink[s] = snk; he_corner[s] = c; break; }
      } else {
        t.opp_own[c] = found;
        t.opp_own[found] = c;
      }
    }
    t.V = nv;
  }

  // ---- the same tables on host threads, for the inputs whose result does not depend on the corner order ----
  // With no vertex-degenerate face and no undirected edge shared by more than two faces, the bucket matching above links corner c to
  // the one corner c' that carries the reverse half-edge (sink → source) unless both have the same tip (Q22) — whichever of the two
  // comes first — so the table can be built from complete buckets in any order.  The edge count is has_non_manifold_edge()'s
  // predicate (≥ 3 faces on an edge): such meshes, and meshes with degenerate faces, return false and take the serial path.
  // kThreads = false: the same passes on the calling thread with plain adds (a small mesh of a batch: no sort, no deletions, and the
  // non-manifold-edge test for free — faster than the literal walk + has_non_manifold_edge()).
  template <bool kThreads>
  bool match_half_edges_parallel() {
    const uint32_t nv = t.V;
    auto bump = [](uint32_t* p) -> uint32_t { if (kThreads) return __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); return (*p)++; };
    auto slices = [&](size_t n, auto&& fn) { if (kThreads) parallel_for(n, fn); else fn((size_t)0, n); };
    // Half-edges are bucketed by the SMALLER endpoint of their undirected edge, tagged with their direction: one scan of one bucket then
    // shows a corner both the half-edges that run its way and the ones that run against it.
    if (nv >= (1u << 31)) return false;   // (the tag lives in bit 31)
    Pooled<uint32_t> count_p((size_t)nv + 1, 0u);
    std::vector<uint32_t>& count = count_p.v;
    std::atomic<int> degenerate{0};
    slices(t.F, [&](size_t lo, size_t hi) {
      for (size_t f = lo; f < hi; ++f) {
        const uint32_t a = t.c2v_own[3 * f], b = t.c2v_own[3 * f + 1], c = t.c2v_own[3 * f + 2];
        if (a == b || b == c || a == c) { degenerate.store(1, std::memory_order_relaxed); return; }
        bump(&count[std::min(a, b)]);
        bump(&count[std::min(b, c)]);
        bump(&count[std::min(c, a)]);
      }
    });
    if (degenerate.load()) return false;
    Pooled<uint32_t> start_p((size_t)nv + 1, 0u), cursor_p((size_t)nv), he_key_p(C), he_corner_p(C);
    std::vector<uint32_t>&start = start_p.v, &cursor = cursor_p.v, &he_key = he_key_p.v, &he_corner = he_corner_p.v;
    for (uint32_t v = 0; v < nv; ++v) start[v + 1] = start[v] + count[v];
    cursor.assign(start.begin(), start.end() - 1);
    he_key.resize(C); he_corner.resize(C);
    slices(C, [&](size_t lo, size_t hi) {
      for (size_t c = lo; c < hi; ++c) {
        const uint32_t src = t.c2v_own[corner_next((uint32_t)c)], snk = t.c2v_own[corner_prev((uint32_t)c)];
        const uint32_t slot = bump(&cursor[std::min(src, snk)]);
        he_key[slot] = src < snk ? snk : (src | 0x80000000u);   // the larger endpoint; bit 31: the half-edge runs from it down
        he_corner[slot] = (uint32_t)c;
      }
    });
    pool_fit(t.opp_own, C);
    t.opp_own.resize(C);
    std::atomic<int> crowded{0};
    slices(C, [&](size_t lo, size_t hi) {
      for (size_t c = lo; c < hi; ++c) {
        const uint32_t tip = t.c2v_own[c], src = t.c2v_own[corner_next((uint32_t)c)], snk = t.c2v_own[corner_prev((uint32_t)c)];
        const uint32_t low = std::min(src, snk);
        const uint32_t mine = src < snk ? snk : (src | 0x80000000u), against = mine ^ 0x80000000u;
        uint32_t same = 0, rev = 0, found = kNone;
        for (uint32_t s2 = start[low]; s2 < start[low + 1]; ++s2) {
          const uint32_t k = he_key[s2];
          same += k == mine;
          if (k == against) { ++rev; found = he_corner[s2]; }
        }
        if (same + rev > 2) { crowded.store(1, std::memory_order_relaxed); return; }
        t.opp_own[c] = (rev == 1 && same == 1 && t.c2v_own[found] != tip) ? found : kNone;
      }
    });
    return !crowded.load();
  }

  // Left-most corners when every vertex has ONE fan (no vertex is split, mod.rs:368-385): for an open fan the left-most corner is
  // where swinging left ends whatever the start; for a closed fan the serial walk starts at the vertex's first corner c in corner
  // order and stops on the corner before c — swing_right(c).  A vertex whose fan does not hold all of its corners has several
  // fans: false, and the serial walk (which splits such vertices) runs instead.
  bool left_most_corners_parallel() {
    const uint32_t nv = t.V;
    Pooled<uint32_t> first_p(nv, kNone), count_p(nv, 0u);
    std::vector<uint32_t>&first = first_p.v, &count = count_p.v;
    parallel_for(C, [&](size_t lo, size_t hi) {
      for (size_t c = lo; c < hi; ++c) {
        const uint32_t v = t.c2v_own[c];
        __atomic_fetch_add(&count[v], 1u, __ATOMIC_RELAXED);
        uint32_t cur = __atomic_load_n(&first[v], __ATOMIC_RELAXED);
        while ((uint32_t)c < cur && !__atomic_compare_exchange_n(&first[v], &cur, (uint32_t)c, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
      }
    });
    pool_fit(t.lmc_own, nv);
    t.lmc_own.assign(nv, kNone);
    std::atomic<int> several{0};
    parallel_for(nv, [&](size_t lo, size_t hi) {
      for (size_t v = lo; v < hi; ++v) {
        const uint32_t c = first[v];
        if (c == kNone) continue;
        uint32_t fan = 1, left = c, a = swing_left(c);
        while (a != kNone && a != c && fan <= count[v]) { left = a; ++fan; a = swing_left(a); }
        if (a == kNone) for (uint32_t r = swing_right(c); r != kNone && fan <= count[v]; r = swing_right(r)) ++fan;   // open fan: the corners to the right of c
        if (fan != count[v]) { several.store(1, std::memory_order_relaxed); return; }
        t.lmc_own[v] = left;
      }
    });
    return !several.load();
  }

  // "some undirected edge has more than two faces" (mod.rs:121-145) without the global sort:
  // group edges by their smaller endpoint (counting sort), then sort each small group.
  bool has_non_manifold_edge() const {
    const uint32_t nv = t.V;
    std::vector<uint32_t> start(nv + 1, 0);
    for (uint32_t f = 0; f < t.F; ++f)
      for (int k = 0; k < 3; ++k) {
        uint32_t a = t.c2v_own[3 * f + k], b = t.c2v_own[3 * f + (k + 1) % 3];
        ++start[std::min(a, b) + 1];
      }
    for (uint32_t v = 0; v < nv; ++v) start[v + 1] += start[v];
    std::vector<uint32_t> other(C), fill(start.begin(), start.end() - 1);
    for (uint32_t f = 0; f < t.F; ++f)
      for (int k = 0; k < 3; ++k) {
        uint32_t a = t.c2v_own[3 * f + k], b = t.c2v_own[3 * f + (k + 1) % 3];
        other[fill[std::min(a, b)]++] = std::max(a, b);
      }
    for (uint32_t v = 0; v < nv; ++v) {
      uint32_t* lo = other.data() + start[v];
      uint32_t* hi = other.data() + start[v + 1];
      if (hi - lo < 3) continue;
      std::sort(lo, hi);
      for (uint32_t* p = lo + 2; p < hi; ++p) if (p[0] == p[-1] && p[0] == p[-2]) return true;
    }
    return false;
  }

  // Break connectivity at non-manifold edges (mod.rs:149-234, following Draco).
  void break_non_manifold_edges() {
    std::vector<uint8_t> seen(C, 0);
    std::vector<std::pair<uint32_t, uint32_t>> sinks;
    bool changed;
    do {
      changed = false;
      for (uint32_t c0 = 0; c0 < C; ++c0) {
        if (seen[c0]) continue;
        sinks.clear();
        uint32_t first = c0, cur = c0;
        for (uint32_t n; (n = swing_left(cur)) != kNone && n != first && !seen[n];) cur = n;
        first = cur;
        for (;;) {
          seen[cur] = 1;
          const uint32_t sink_c = corner_next(cur), sink_v = t.c2v_own[sink_c], edge_c = corner_prev(cur);
          bool updated = false;
          for (auto& s : sinks) {
            if (s.first != sink_v) continue;
            const uint32_t other_edge = s.second, oe = t.opp_own[edge_c];
            if (oe != kNone && oe == other_edge) continue;
            const uint32_t oo = t.opp_own[other_edge];
            if (oe != kNone) t.opp_own[oe] = kNone;
            if (oo != kNone) t.opp_own[oo] = kNone;
            t.opp_own[edge_c] = kNone;
            t.opp_own[other_edge] = kNone;
            updated = true;
            break;
          }
          if (updated) { changed = true; break; }
          sinks.emplace_back(t.c2v_own[corner_prev(cur)], sink_c);
          const uint32_t r = swing_right(cur);
          if (r == kNone) break;
          cur = r;
          if (cur == first) break;
        }
      }
    } while (changed);
  }

  // Left-most corners + non-manifold vertex splitting (mod.rs:342-416).
  void left_most_corners() {
    t.lmc_own.assign(t.V, kNone);
    std::vector<uint8_t> vdone(t.V, 0), cdone(C, 0);
    for (uint32_t c = 0; c < C; ++c) {
      if (cdone[c]) continue;
      uint32_t v = t.c2v_own[c];
      const bool split = vdone[v] != 0;
      if (split) { v = t.V++; t.lmc_own.push_back(kNone); vdone.push_back(0); }
      vdone[v] = 1;
      cdone[c] = 1;
      t.lmc_own[v] = c;
      if (split) t.c2v_own[c] = v;
      uint32_t a = swing_left(c);
      while (a != kNone && a != c) {
        cdone[a] = 1;
        t.lmc_own[v] = a;
        if (split) t.c2v_own[a] = v;
        a = swing_left(a);
      }
      if (a == kNone) {
        for (a = c; a != kNone; a = swing_right(a)) { cdone[a] = 1; if (split) t.c2v_own[a] = v; }
      }
    }
  }
};

}  // namespace

int CornerTables::build_universal(const uint32_t* faces, uint32_t num_faces, const uint32_t* pos_p2v, std::string& err, bool copy_faces) {
  F = num_faces;
  const uint32_t C = 3 * F;
  const bool trace = dbg_on(DMI_DBG_TRACE_TABLES);
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  if (copy_faces) { pool_fit(c2p_own, C); c2p_own.resize(C); parallel_for(C, [&](size_t lo, size_t hi) { std::copy(faces + lo, faces + hi, c2p_own.data() + lo); }); c2p = c2p_own.data(); }
  else c2p = faces;   // (the caller's array outlives these tables: dmi_mesh_prepare / dmi_encode_mesh)
  pool_fit(c2v_own, C);
  c2v_own.resize(C);
  std::vector<uint32_t>& c2v = c2v_own;
  std::atomic<uint32_t> maxv_a{0};
  parallel_for(C, [&](size_t lo, size_t hi) {
    uint32_t m = 0;
    for (size_t c = lo; c < hi; ++c) { c2v[c] = pos_p2v ? pos_p2v[faces[c]] : faces[c]; m = std::max(m, c2v[c]); }
    uint32_t cur = maxv_a.load();
    while (m > cur && !maxv_a.compare_exchange_weak(cur, m)) {}
  });
  V = C ? maxv_a.load() + 1 : 0;
  {   // core/corner_table/mod.rs:105-108: unused vertex ids are a panic in the reference
    Pooled<uint8_t> used_p(V, (uint8_t)0);
    std::vector<uint8_t>& used = used_p.v;
    parallel_for(C, [&](size_t lo, size_t hi) { for (size_t c = lo; c < hi; ++c) used[c2v[c]] = 1; });   // (racing stores of the same value)
    std::atomic<int> unused{0};
    parallel_for(V, [&](size_t lo, size_t hi) { for (size_t v = lo; v < hi; ++v) if (!used[v]) { unused.store(1); break; } });
    if (unused.load()) { err = "mesh contains unused vertices"; return DMI_ERR_UNUSED_VERTICES; }
  }
  const double t_ids = ms();
  UniversalBuilder b(*this);
  const bool serial_only = dbg_on(DMI_DBG_SERIAL_TABLES);   // (tests: the literal serial walks on every input)
  // (the order-independent builders do more work per corner — atomics, two bucket scans — and only win once their loops really run on
  //  several threads, which parallel_for does from 2^20 items; below that, and in a batch of meshes on a thread each, the serial walks)
  const bool big = (C >= (1u << 21) || dbg_on(DMI_DBG_PARALLEL_TABLES)) && !serial_only;
  const bool matched = serial_only ? false : (big ? b.match_half_edges_parallel<true>() : b.match_half_edges_parallel<false>());
  if (!matched) {
    b.match_half_edges();
    if (b.has_non_manifold_edge()) b.break_non_manifold_edges();
  }
  const double t_match = ms();
  if (!(big && b.left_most_corners_parallel())) b.left_most_corners();
  if (trace) std::fprintf(stderr, "[dmi]   universal table of %u faces: copy + vertex ids %.3f ms, half-edge matching %.3f (%s), left-most corners %.3f\n", F, t_ids, t_match - t_ids,
                          matched ? "order-free" : "reference walk", ms() - t_match);
  this->c2v = c2v_own.data(); opp = opp_own.data(); lmc = lmc_own.data();
  no_boundary = false;
  att.clear();
  return DMI_OK;
}

// core/corner_table/attribute_corner_table.rs:16-137
void CornerTables::build_attribute(const uint32_t* p2v) {
  att.emplace_back();
  build_attribute_into(att.back(), p2v);
}

// (reads the universal table only: attribute tables of one mesh can be built concurrently; the loops of a large mesh run on host
// threads themselves — seam flags are idempotent byte stores, the attribute-vertex ids of a universal vertex are a prefix sum over
// the per-vertex counts: exactly the ids the serial `nv++` walk hands out)
void CornerTables::copy_attribute_into(AttTable& a, const AttTable& from) const {
  auto copy = [](auto& dst, const auto& src) { pool_fit(dst, src.size()); dst.assign(src.begin(), src.end()); };
  copy(a.seam_edge, from.seam_edge); copy(a.c2v, from.c2v); copy(a.opp, from.opp); copy(a.lmc, from.lmc);
  a.num_vertices = from.num_vertices;
  a.interior_seams = from.interior_seams;
}

void CornerTables::build_attribute_into(AttTable& a, const uint32_t* p2v, bool same_as_position) const {
  const uint32_t C = 3 * F;
  a.interior_seams = false;
  a.alias_of = -1;
  pool_give(a.c2v); pool_give(a.opp); pool_give(a.lmc);
  a.num_vertices = V;
  if (same_as_position) {
    // the universal vertices ARE this attribute's values: only the boundary edges are seams.  Nobody reads the flags of such an attribute (its
    // seam stream is all zeros — coded by its period — and its table is the universal one): they are not materialised (6 bytes of writes per
    // face and attribute pair in a batch of seam-free meshes)
    pool_give(a.seam_edge);
    return;
  }
  pool_fit(a.seam_edge, C);
  a.seam_edge.assign(C, 0);
  Pooled<uint8_t> vseam_p(V, (uint8_t)0);
  std::vector<uint8_t>& vseam = vseam_p.v;
  std::atomic<int> interior{0};
  parallel_for(C, [&](size_t lo, size_t hi) {
    bool any = false;
    struct Note { std::atomic<int>& f; bool& any; ~Note() { if (any) f.store(1, std::memory_order_relaxed); } } note{interior, any};
    for (size_t cc = lo; cc < hi; ++cc) {
      const uint32_t c = (uint32_t)cc;
      const uint32_t o = opp[c];
      if (o == kNone) {
        a.seam_edge[c] = 1;
        vseam[c2v[corner_next(c)]] = 1;
        vseam[c2v[corner_prev(c)]] = 1;
        continue;
      }
      if (o < c) continue;
      // the two shared endpoints: next(c)↔prev(o) and prev(c)↔next(o) — the same POINT on both sides (the rule inside an indexed mesh) needs no
      // look-up of its value
      const uint32_t pa = c2p[corner_next(c)], pb = c2p[corner_prev(o)], pc = c2p[corner_prev(c)], pd = c2p[corner_next(o)];
      auto value_of = [&](uint32_t p) { return p2v ? p2v[p] : p; };
      if ((pa != pb && value_of(pa) != value_of(pb)) || (pc != pd && value_of(pc) != value_of(pd))) {
        a.seam_edge[c] = a.seam_edge[o] = 1;
        vseam[c2v[corner_next(c)]] = vseam[c2v[corner_prev(c)]] = 1;
        vseam[c2v[corner_next(o)]] = vseam[c2v[corner_prev(o)]] = 1;
        any = true;
      }
    }
  });
  a.interior_seams = interior.load() != 0;
  if (!a.interior_seams) return;   // (no table of its own: every consumer takes the universal one)
  finish_attribute(a, vseam);
}

// The attribute table of a decoder: the seam flags come from the bitstream (DefaultTraversal's seam stream, edgebreaker.rs:611-653) instead of
// from value comparisons; a.seam_edge must hold them for BOTH corners of every seam edge and for every boundary corner.
void CornerTables::attribute_from_seams(AttTable& a) const {
  const uint32_t C = 3 * F;
  a.alias_of = -1;
  a.num_vertices = V;
  a.interior_seams = false;
  std::vector<uint8_t> vseam(V, 0);
  for (uint32_t c = 0; c < C; ++c) {
    if (!a.seam_edge[c]) continue;
    vseam[c2v[corner_next(c)]] = vseam[c2v[corner_prev(c)]] = 1;
    if (opp[c] != kNone) a.interior_seams = true;
  }
  if (a.interior_seams) finish_attribute(a, vseam);
}

// attribute_corner_table.rs:79-137 (recompute_vertices) from the seam flags
void CornerTables::finish_attribute(AttTable& a, const std::vector<uint8_t>& vseam) const {
  const uint32_t C = 3 * F;
  pool_fit(a.opp, C);
  a.opp.resize(C);
  parallel_for(C, [&](size_t lo, size_t hi) { for (size_t c = lo; c < hi; ++c) a.opp[c] = a.seam_edge[c] ? kNone : opp[c]; });
  pool_fit(a.c2v, C);
  a.c2v.assign(C, 0);
  auto a_swing_left = [&](uint32_t c) { uint32_t o = a.opp[corner_next(c)]; return o == kNone ? kNone : corner_next(o); };
  auto u_swing_right = [&](uint32_t c) { uint32_t o = opp[corner_prev(c)]; return o == kNone ? kNone : corner_prev(o); };
  auto fan_start = [&](uint32_t v) {   // seam-aware swing to the fan start (attribute_corner_table.rs:101-113)
    uint32_t first = lmc[v];
    if (vseam[v]) for (uint32_t n; (n = a_swing_left(first)) != kNone && n != lmc[v];) first = n;
    return first;
  };
  // attribute vertices per universal vertex: 1 + the seam edges its right swing crosses (:116-133)
  Pooled<uint32_t> base_p((size_t)V + 1, 0u);
  std::vector<uint32_t>& base = base_p.v;
  parallel_for(V, [&](size_t lo, size_t hi) {
    for (size_t v = lo; v < hi; ++v) {
      uint32_t k = 1;
      if (vseam[v]) { const uint32_t first = fan_start((uint32_t)v); for (uint32_t cur = u_swing_right(first); cur != kNone && cur != first; cur = u_swing_right(cur)) k += a.seam_edge[corner_next(cur)]; }
      base[v + 1] = k;
    }
  });
  for (uint32_t v = 0; v < V; ++v) base[v + 1] += base[v];
  const uint32_t nv = base[V];
  pool_fit(a.lmc, nv);
  a.lmc.assign(nv, kNone);
  // a vertex no seam touches keeps ONE attribute vertex, its fan start is the universal left-most corner: its corners take their id in a
  // streaming pass over the corners — only the vertices ON a seam (a vanishing share of a mesh) walk their fans
  parallel_for(C, [&](size_t lo, size_t hi) { for (size_t c = lo; c < hi; ++c) { const uint32_t v = c2v[c]; if (!vseam[v]) a.c2v[c] = base[v]; } });
  parallel_for(V, [&](size_t lo, size_t hi) {
    for (size_t v = lo; v < hi; ++v) {
      if (!vseam[v]) { a.lmc[base[v]] = lmc[v]; continue; }
      const uint32_t first = fan_start((uint32_t)v);
      uint32_t id = base[v];
      a.c2v[first] = id;
      a.lmc[id] = first;
      for (uint32_t cur = u_swing_right(first); cur != kNone && cur != first; cur = u_swing_right(cur)) {
        if (a.seam_edge[corner_next(cur)]) { ++id; a.lmc[id] = cur; }
        a.c2v[cur] = id;
      }
    }
  });
  a.num_vertices = nv;
}

}  // namespace dmi
