// host_conn.cpp — corner tables, Edgebreaker traversal + connectivity bytes, attribute sequencer.
// CPU, single thread per mesh, flat arrays.  Output is bit-identical to the reference's
// (quirks kept: SURVEY.md §8a-Q Q22 half-edge matching, per-vertex "hole" ids).
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
WalkSlots& walk_slots() { static WalkSlots w; return w; }


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
  static const bool trace = std::getenv("DMI_TRACE_TABLES") != nullptr;
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
  const bool serial_only = std::getenv("DMI_SERIAL_TABLES") != nullptr;   // (tests: the literal serial walks on every input)
  // (the order-independent builders do more work per corner — atomics, two bucket scans — and only win once their loops really run on
  //  several threads, which parallel_for does from 2^20 items; below that, and in a batch of meshes on a thread each, the serial walks)
  const bool big = (C >= (1u << 21) || std::getenv("DMI_PARALLEL_TABLES")) && !serial_only;
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

// ------------------------------------------------------------------------------------------------
// Edgebreaker (encode/connectivity/edgebreaker.rs), Standard traversal.
// ------------------------------------------------------------------------------------------------
namespace {
enum : uint8_t { SYM_C, SYM_S, SYM_L, SYM_R, SYM_E };
// set bits of a three-bit edge mask (the generic x86-64 target has no popcnt: __builtin_popcount is a dozen instructions — 2 ns per face of a batch's seam stage)
static const uint8_t kBits3[8] = {0, 1, 1, 2, 1, 2, 2, 3};
// Adjacent-line prefetch for the serial walks.  An Edgebreaker spiral (and the sequencer's depth-first walk) sweeps a front across the
// mesh: the entries a loop of the front touches lie next to — in memory: in the same or the neighbouring cache line of — the entries the
// previous loop touched.  Touching line L therefore asks for L−1 and L+1 as well: when the front reaches them, a loop later, they wait in
// L2 instead of DRAM (10M-triangle grid on the GPU box's EPYC: traversal 144 → 93 ms, sequencer 121 → 64 ms together with huge pages).
// The distance is a constant of the loops (the neighbouring 64-byte line on either side): as a run-time setting (rounds 3–4: DMI_PF=<entries>) it cost
// every step a load, a test and the address arithmetic of each prefetch — a dozen of a step's ≈ 75 instructions, in loops bound by instruction issue.
// The flag arrays stay cache-resident by themselves (prefetching their neighbour lines cost the lean loops 5 %: not done).
inline void prefetch_neighbours(const uint32_t* p) { __builtin_prefetch(p + 16, 0, 2); __builtin_prefetch(p - 16, 0, 2); }
inline void prefetch_neighbours(const uint8_t*) {}

// The size of a vector of trivial elements whose first n slots (n ≤ capacity) were written through data(): the walks fill their output arrays
// (one entry per face / vertex, capacity known up front) through raw pointers — a push_back per step is a call the compilers do not inline
// (g++: emplace_back<unsigned char>; clang: the split bookkeeping) and cost the 10M-face traversal a third of its time — and resize() would
// zero-fill 50 MB first.  libstdc++ keeps {begin, end, end of storage}; a layout that does not answer the probe takes the plain resize + copy.
template <class T>
inline void set_size_written(std::vector<T>& v, size_t n) {
  static_assert(std::is_trivially_copyable<T>::value, "trivial elements only");
  struct Impl { T* b; T* e; T* c; };
  if (sizeof(std::vector<T>) == sizeof(Impl) && n <= v.capacity()) {
    Impl& im = reinterpret_cast<Impl&>(v);
    if (im.b == v.data() && im.e == v.data() + v.size() && im.c == v.data() + v.capacity()) { im.e = im.b + n; return; }
  }
  std::vector<T> copy(v.data(), v.data() + n);   // (never taken with libstdc++)
  v.swap(copy);
}

// Corner ids inside the two serial walks.  A step goes corner → face → position in the face → next corner → opposite[next] → …: with ids
// 3·face + k the face is a division (multiply-high + shift ON the dependency chain of the walk) and k a multiply-subtract; with ids 4·face + k (Enc4) a
// shift and a mask.  The arrays stay dense — three entries per face, entry of corner c at c − face — only the VALUES of `opposite` (CornerTables::quad,
// written that way by the device stage, dmi_conn.hip k_opp_quad) and what a walk keeps on its stack / in `processed` are 4·face + k.  10M-face grid on the
// GPU box's EPYC, traversal loop alone: 3.88 → 3.59 ns per face (scripts/experiments/walk_enc4.cpp).
struct Enc3 {
  static constexpr bool kQuad = false;
  static uint32_t face(uint32_t c) { return c / 3; }
  static uint32_t k(uint32_t c, uint32_t f) { return c - 3 * f; }
  static uint32_t idx(uint32_t c, uint32_t) { return c; }            // where corner c's entries are
  static uint32_t from3(uint32_t c3) { return c3; }
  static uint32_t to3(uint32_t c) { return c; }
  static uint32_t dense(uint32_t c) { return c; }                     // to3 of a corner known to exist
};
struct Enc4 {
  static constexpr bool kQuad = true;
  static uint32_t face(uint32_t c) { return c >> 2; }
  static uint32_t k(uint32_t c, uint32_t) { return c & 3u; }
  static uint32_t idx(uint32_t c, uint32_t f) { return c - f; }
  static uint32_t from3(uint32_t c3) { return c3 == kNone ? kNone : c3 + c3 / 3; }
  static uint32_t to3(uint32_t c) { return c == kNone ? kNone : c - (c >> 2); }
  static uint32_t dense(uint32_t c) { return c - (c >> 2); }
};

struct Walker {
  const CornerTables& t;
  const uint32_t C;
  const bool quad;                    // t.opp holds 4·face + k ids (the walk then keeps its stack and `processed` in that form: EdgebreakerResult::processed_quad)
  // opposite corner as a 3·face + k id whatever the table holds (everything outside the traversal loop)
  uint32_t opp3(uint32_t c3) const { const uint32_t o = t.opp[c3]; return quad ? Enc4::to3(o) : o; }
  std::vector<uint8_t> vvis, fvis, hole_done;   // vvis: bit 0 visited, bit 1 the vertex lies on a boundary (hole_of != kNone); fvis: bit 0 visited, bit 1 an S face
  std::vector<uint32_t> hole_of;      // per vertex, kNone = interior
  std::vector<uint32_t> stack, processed, init_corners;
  std::unordered_map<uint32_t, uint64_t> split_symbol_of_face;   // S faces only (flagged in fvis bit 1)
  std::vector<uint8_t> symbols, start_interior;
  struct Split { uint64_t merging, split; uint8_t right; };
  std::vector<Split> splits;
  uint64_t num_split_symbols = 0;
  size_t n_out = 0;                   // faces processed so far = symbols written (processed / symbols are filled through data(): finish() sets their size)
  bool bad = false;

  // Meshes of ≥ 2^16 faces keep their face flags as 32-bit STAMPS (0 = unvisited, else position in `processed` + 1; bit 31: an S face): a step that sees a visited
  // neighbour then knows where in `processed` the spiral's previous loop passed this spot, a shadow index follows the walk one loop behind, and the
  // table lines of the face that loop processed a dozen steps later — the neighbours of what this walk reaches a dozen steps from now — are
  // requested into L1: the hop of a step (corner → opposite[next(corner)] → next corner) then hits L1 instead of L2 (traversal of the 10M-triangle
  // grid −7…13 % in the stand-alone loop, scripts/experiments/walk_layout.cpp).  Smaller meshes keep byte flags (their tables sit in L2).
  std::vector<uint32_t> stamp;
  bool use_stamp = false;
  static constexpr uint32_t kStampS = 0x80000000u, kStampStart = 0x7FFFFFFFu;
  bool face_visited(uint32_t f) const { return use_stamp ? stamp[f] != 0u : (fvis[f] & 1) != 0; }
  void mark_start_face(uint32_t f) { if (use_stamp) stamp[f] = kStampStart; else fvis[f] |= 1; }

  explicit Walker(const CornerTables& tt) : t(tt), C(tt.F * 3), quad(tt.quad) {
    pool_fit(vvis, t.V); vvis.assign(t.V, 0);
    static const bool no_shadow = std::getenv("DMI_NO_SHADOW") != nullptr;
    static const uint32_t min_faces = std::getenv("DMI_SHADOW_MIN_FACES") ? (uint32_t)std::atol(std::getenv("DMI_SHADOW_MIN_FACES")) : (1u << 16);   // (256-mesh batch: traversal thread time 71 → 68 ms with the meshes of ≥ 2^16 faces on stamps; below, tables and flags sit in L2)
    use_stamp = t.F >= min_faces && t.F < 0x7FFFFFF0u && !no_shadow;
    if (use_stamp) { pool_fit(stamp, t.F); stamp.assign(t.F, 0u); }
    else { pool_fit(fvis, t.F); fvis.assign(t.F, 0); }
    pool_fit(processed, (size_t)t.F + 1); if (processed.capacity() < (size_t)t.F + 1) processed.reserve((size_t)t.F + 1);
    pool_fit(symbols, (size_t)t.F + 1); if (symbols.capacity() < (size_t)t.F + 1) symbols.reserve((size_t)t.F + 1);
  }
  ~Walker() { pool_give(vvis); pool_give(fvis); pool_give(stamp); pool_give(hole_of); pool_give(processed); pool_give(symbols); }
  Walker(const Walker&) = delete;
  Walker& operator=(const Walker&) = delete;
  void finish() { set_size_written(processed, n_out); set_size_written(symbols, n_out); }
  uint32_t swing_right(uint32_t c) const { uint32_t o = opp3(corner_prev(c)); return o == kNone ? kNone : corner_prev(o); }

  // edgebreaker.rs:195-224 — the inner walk rotates inside one face (never crosses an edge), so
  // each boundary vertex ends up with its own id.
  void label_boundaries() {
    if (t.no_boundary) return;   // (no vertex flag is ever set: hole_of is never read)
    pool_fit(hole_of, t.V);
    hole_of.assign(t.V, kNone);
    for (uint32_t c0 = 0; c0 < C; ++c0) {
      if (t.opp[c0] != kNone) continue;
      uint32_t v = t.c2v[corner_next(c0)];
      if (hole_of[v] != kNone) continue;
      const uint32_t id = (uint32_t)hole_done.size();
      hole_done.push_back(0);
      uint32_t c = c0;
      while (hole_of[v] == kNone) {
        hole_of[v] = id;
        vvis[v] |= 2;
        c = corner_next(c);
        while (t.opp[c] != kNone) c = corner_next(c);
        v = t.c2v[corner_next(c)];
      }
    }
  }
  // edgebreaker.rs:226-256
  void mark_boundary(uint32_t start_corner, bool include_first) {
    uint32_t c = corner_prev(start_corner);
    while (t.opp[c] != kNone) c = corner_next(opp3(c));
    const uint32_t sv = t.c2v[start_corner];
    if (include_first) vvis[sv] |= 1;
    if (hole_of.empty() || hole_of[sv] == kNone) { bad = true; return; }
    hole_done[hole_of[sv]] = 1;
    for (uint32_t v = t.c2v[corner_prev(c)]; v != sv; v = t.c2v[corner_prev(c)]) {
      vvis[v] |= 1;
      c = corner_next(c);
      while (t.opp[c] != kNone) c = corner_next(opp3(c));
    }
  }
  // (rare: the face across the edge is an S face — the 8-byte-per-face array of the reference is a map over the S faces only)
  __attribute__((noinline)) void note_split(uint64_t merging, uint8_t right, uint32_t face) { splits.push_back({merging, split_symbol_of_face[face], right}); }
  // an S face: the left branch waits on the stack, the right one is walked first (rare: kept out of the loop's registers)
  // (c3: the corner as a 3·face + k id; rc / lc: as the loop keeps them)
  __attribute__((noinline)) void split_here(uint32_t c3, uint32_t f, uint32_t v, uint8_t vflags, uint32_t rc, uint32_t lc, uint64_t symbol_idx) {
    ++num_split_symbols;
    if ((vflags & 2) && !hole_done[hole_of[v]]) mark_boundary(c3, false);
    split_symbol_of_face[f] = symbol_idx;
    if (use_stamp) stamp[f] |= kStampS; else fvis[f] |= 2;
    stack.back() = lc;
    stack.push_back(rc);
  }
  // edgebreaker.rs:261-350.  The loop keeps its tables and outputs in locals (every flag store is a byte store, which may alias anything the
  // object holds: members would be reloaded after each of them) and writes one (corner, symbol) pair per step through raw pointers.
  // c3: the corner to start from, a 3·face + k id
  void run_from(uint32_t c3) {
    // (kClosed: every corner has an opposite — the device stage reports it — so the loop tests no entry for "none")
    static const bool no_closed = std::getenv("DMI_NO_CLOSED") != nullptr;
    const bool closed = t.no_boundary && !no_closed;
    if (quad) {
      if (use_stamp) { if (closed) run_from_t<true, Enc4, true>(Enc4::from3(c3)); else run_from_t<true, Enc4, false>(Enc4::from3(c3)); }
      else run_from_t<false, Enc4, false>(Enc4::from3(c3));
    } else {
      if (use_stamp) { if (closed) run_from_t<true, Enc3, true>(c3); else run_from_t<true, Enc3, false>(c3); }
      else run_from_t<false, Enc3, false>(c3);
    }
  }
  template <bool kStamp, class E, bool kClosed>
  void run_from_t(const uint32_t c_start) {
    const uint32_t* const opp = t.opp;
    const uint32_t* const c2v = t.c2v;
    uint8_t* const fv = fvis.data();
    uint32_t* const st = stamp.data();
    uint8_t* const vv = vvis.data();
    uint32_t* const proc = processed.data();
    uint8_t* const sym = symbols.data();
    const size_t cap = t.F;            // a consistent table never processes a face twice: more symbols than faces ⇒ malformed (the reference would not terminate)
    size_t n = n_out;
    size_t q = ~(size_t)0 >> 1;        // the shadow: position in `processed` of the previous loop's face beside this one (far away until a step sees a visited neighbour)
    constexpr size_t kAhead = 12;
    stack.clear();
    stack.push_back(c_start);
    uint32_t c;                        // (a local of its own: the parameter's address is taken by push_back — it would live in memory, stored every step)
    while (!stack.empty() && !bad) {
      c = stack.back();
      if (c == kNone) { bad = true; break; }
      if (kStamp ? st[E::face(c)] != 0u : (fv[E::face(c)] & 1) != 0) { stack.pop_back(); continue; }
      for (;;) {
        if ((!kClosed && c == kNone) || n >= cap) { bad = true; break; }
        const uint32_t f = E::face(c), k = E::k(c, f), i = E::idx(c, f);
        prefetch_neighbours(opp + i); prefetch_neighbours(c2v + i);
        const uint32_t v = c2v[i];
        const uint32_t cn = k == 2 ? c - 2 : c + 1;
        if (kStamp) {
          const size_t qa = q + kAhead;
          if (qa < n) { const uint32_t g = E::dense(proc[qa]); __builtin_prefetch(opp + g, 0, 3); __builtin_prefetch(opp + g + 16, 0, 3); __builtin_prefetch(opp + g - 16, 0, 3); __builtin_prefetch(c2v + g, 0, 3); }
          ++q;
          st[f] = (uint32_t)n + 1u;
        } else {
          prefetch_neighbours(fv + f);
          fv[f] |= 1;
        }
        prefetch_neighbours(vv + v);
        proc[n] = c;
        const uint32_t gate = (kClosed || opp[i] != kNone) ? 0x10u : 0u;   // (the face this one was entered from — or a start face — is always visited)
        const uint8_t vflags = vv[v];
        if (!(vflags & 1)) {
          vv[v] = vflags | 1;
          // (a C face: its tip was unvisited, so neither the right nor the left face — both hold the tip — has been processed)
          if (!(vflags & 2)) { sym[n++] = (uint8_t)(SYM_C | gate); c = opp[E::idx(cn, f)]; continue; }
        }
        const uint32_t cp = k == 0 ? c + 2 : c - 1;
        const uint32_t rc = opp[E::idx(cn, f)], lc = opp[E::idx(cp, f)];
        // (neighbour states: byte flags — bit 0 visited, bit 1 S face — or stamps)
        uint32_t rs = 0, ls = 0;
        bool rv, lv, r_split, l_split;
        const bool r_none = !kClosed && rc == kNone, l_none = !kClosed && lc == kNone;
        if (kStamp) {
          rs = r_none ? 0u : st[E::face(rc)]; ls = l_none ? 0u : st[E::face(lc)];
          rv = r_none || rs != 0u; lv = l_none || ls != 0u;
          r_split = (rs & kStampS) != 0u; l_split = (ls & kStampS) != 0u;
        } else {
          const uint8_t rf = rc == kNone ? 1 : fv[E::face(rc)], lf = lc == kNone ? 1 : fv[E::face(lc)];
          rv = rf & 1; lv = lf & 1;
          r_split = rc != kNone && (rf & 2); l_split = lc != kNone && (lf & 2);
        }
        // bits 4–6 of a symbol: which of the edges opposite (c, next, prev) lead to a face processed EARLIER (or to a start face) — what the seam
        // streams emit for this face (edgebreaker.rs:611-636 walks the faces last to first and emits the edges whose other face is not visited yet)
        const uint8_t nb = (uint8_t)(gate | ((!r_none && rv) ? 0x20u : 0u) | ((!l_none && lv) ? 0x40u : 0u));
        const uint64_t symbol_idx = n;   // (symbols so far = the index of this one)
        if (rv) {
          if (kStamp && rs) q = (size_t)(rs & 0x7FFFFFFFu);   // (the right face was processed at position rs - 1: the shadow moves on from the one after it)
          if (r_split) note_split(symbol_idx, 1, E::face(rc));
          if (lv) {
            if (l_split) note_split(symbol_idx, 0, E::face(lc));
            sym[n++] = (uint8_t)(SYM_E | nb);
            stack.pop_back();
            break;
          }
          sym[n++] = (uint8_t)(SYM_R | nb);
          c = lc;
        } else if (lv) {
          if (kStamp && ls) q = (size_t)(ls & 0x7FFFFFFFu);
          if (l_split) note_split(symbol_idx, 0, E::face(lc));
          sym[n++] = (uint8_t)(SYM_L | nb);
          c = rc;
        } else {
          sym[n++] = (uint8_t)(SYM_S | nb);
          split_here(E::dense(c), f, v, vflags, rc, lc, symbol_idx);
          break;
        }
      }
    }
    n_out = n;
  }
  // edgebreaker.rs:411-431
  bool pick_start(uint32_t face, uint32_t& corner) const {
    uint32_t c = 3 * face;
    for (int k = 0; k < 3; ++k) {
      if (t.opp[c] == kNone) { corner = c; return false; }
      if (vvis[t.c2v[c]] & 2) {
        uint32_t r = c;
        while (r != kNone) { c = r; r = swing_right(r); }
        corner = corner_prev(c);
        return false;
      }
      c = corner_next(c);
    }
    corner = c;
    return true;
  }
};
}  // namespace

}  // namespace dmi

// Large heap arrays of THIS library (every std::vector of index / flag arrays: hidden visibility — no other module's allocations come here) start
// on a 2 MiB boundary and end on one, and ask for transparent huge pages as a whole.  malloc hands a 5 MB flag array out 16 bytes into its
// mapping: the 2 MiB-aligned interior that advise_huge_pages can flag leaves its first and last megabytes on 4 KiB pages, and the serial walks
// (one flag byte per step, a mesh row apart: a new page every step) then miss the TLB on 20–40 % of their flag accesses — the 10M-face
// traversal on the GPU box's EPYC: 64 ms against 48 ms with every array on huge pages.  Memory comes from posix_memalign: released by the
// default operator delete (free).  DMI_NO_THP=1: plain malloc.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define DMI_NO_OPERATOR_NEW 1      // (the sanitizer build keeps the runtime's allocator: it pairs operator new with operator delete)
#endif
#endif
#if defined(__SANITIZE_ADDRESS__)
#define DMI_NO_OPERATOR_NEW 1
#endif
#ifndef DMI_NO_OPERATOR_NEW
void* operator new(std::size_t n) {   // (local to the library: libdraco_mi.map)
  constexpr std::size_t kHuge = (std::size_t)2 << 20;
  static const bool off = std::getenv("DMI_NO_THP") != nullptr;
  if (n >= kHuge && !off) {
    const std::size_t want = (n + kHuge - 1) & ~(kHuge - 1);
    void* p = nullptr;
    if (want >= n && posix_memalign(&p, kHuge, want) == 0 && p) { (void)madvise(p, want, MADV_HUGEPAGE); return p; }
  }
  if (void* p = std::malloc(n ? n : 1)) return p;
  throw std::bad_alloc();
}
void* operator new[](std::size_t n) { return ::operator new(n); }
#endif

namespace dmi {
void advise_huge_pages(void* p, size_t bytes) {
  static const bool off = std::getenv("DMI_NO_THP") != nullptr;
  if (off || !p) return;
  constexpr uintptr_t kHuge = (uintptr_t)2 << 20;
  const uintptr_t lo = ((uintptr_t)p + kHuge - 1) & ~(kHuge - 1), hi = ((uintptr_t)p + bytes) & ~(kHuge - 1);
  if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
}

size_t host_pool_limit() {
  static const size_t limit = [] {
    const char* e = std::getenv("DMI_HOST_CACHE_MB");
    return (size_t)(e ? std::max(0l, std::atol(e)) : 4096l) << 20;
  }();
  return limit;
}
std::atomic<size_t>& host_pool_bytes() { static std::atomic<size_t> b{0}; return b; }
void host_pool_drop_all() { VecPool<uint8_t>::get().drop_all(); VecPool<uint32_t>::get().drop_all(); VecPool<uint64_t>::get().drop_all(); }

bool append_tagged_state(uint32_t s, std::vector<uint8_t>& out) {   // rans.rs:48-68
  if (s < (1u << 6)) out.push_back((uint8_t)s);
  else if (s < (1u << 14)) { uint32_t v = (1u << 14) + s; out.push_back((uint8_t)v); out.push_back((uint8_t)(v >> 8)); }
  else if (s < (1u << 22)) { uint32_t v = (2u << 22) + s; out.push_back((uint8_t)v); out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)(v >> 16)); }
  else if (s < (1u << 30)) { uint32_t v = (3u << 30) + s; for (int k = 0; k < 4; ++k) out.push_back((uint8_t)(v >> (8 * k))); }
  else return false;
  return true;
}
bool RabsHost::finish() { return append_tagged_state(state - 4096, out); }

uint8_t zero_probability(uint64_t count_zero, float denominator) {
  float p = ((float)count_zero / denominator) * 256.0f + 0.5f;
  uint32_t q = (p != p || p <= 0.0f) ? 0u : (p >= 65535.0f ? 65535u : (uint32_t)p);   // Rust `as u16`
  return (uint8_t)std::min(255u, std::max(1u, q));
}

std::atomic<uint64_t> g_eb_ns[6];   // trace: thread time of run_edgebreaker by step over all meshes (set-up, traversal, bits, seam streams, count)
int run_edgebreaker(const CornerTables& t, EdgebreakerResult& out, std::string& err, const EdgebreakerHooks* hooks) {
  const bool trace_all = std::getenv("DMI_TRACE") != nullptr;
  struct Whole { bool on; std::chrono::steady_clock::time_point a = std::chrono::steady_clock::now(); ~Whole() { if (on) g_eb_ns[5] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - a).count(); } } whole{trace_all};   // (declared first: its destructor runs after every other local's)
  const bool trace = t.F > 100000 && std::getenv("DMI_TRACE") != nullptr;
  auto tick = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  const auto t0 = tick();
  Walker w(t);
  const double t_ctor = since(t0);
  ByteSink s;
  s.u8(0);   // EdgebreakerKind::Standard
  w.label_boundaries();
  s.leb128(t.V);
  s.leb128(t.F);
  s.u8((uint8_t)t.att.size());
  for (uint32_t f = 0; f < t.F && !w.bad; ++f) {   // edgebreaker.rs:478-511 (loop over corners ≡ loop over faces)
    if (w.face_visited(f)) {
      // (eight visited faces at a time: after the first component's walk this loop only confirms that nothing is left — 10M byte tests otherwise)
      if (w.use_stamp) {
        const uint32_t* st = w.stamp.data();
        while (f + 8 < t.F && st[f + 1] && st[f + 2] && st[f + 3] && st[f + 4] && st[f + 5] && st[f + 6] && st[f + 7] && st[f + 8]) f += 8;
      } else {
        const uint8_t* fv = w.fvis.data();
        while (f + 8 < t.F && (f & 7u) == 7u) {
          uint64_t eight;
          std::memcpy(&eight, fv + f + 1, 8);
          if ((eight & 0x0101010101010101ull) != 0x0101010101010101ull) break;
          f += 8;
        }
      }
      continue;
    }
    uint32_t start;
    const bool interior = w.pick_start(f, start);
    w.start_interior.push_back(interior);
    if (interior) {
      w.vvis[t.c2v[start]] |= 1; w.vvis[t.c2v[corner_next(start)]] |= 1; w.vvis[t.c2v[corner_prev(start)]] |= 1;
      w.mark_start_face(f);
      w.init_corners.push_back(corner_next(start));
      w.run_from(w.opp3(corner_next(start)));
    } else {
      w.mark_boundary(corner_next(start), true);
      w.run_from(start);
    }
  }
  w.finish();
  const double t_walk = since(t0);
  if (trace) std::fprintf(stderr, "[dmi]   (walker set-up %.2f ms)\n", t_ctor);
  if (w.bad) { err = "edgebreaker: inconsistent connectivity (reference unwrap() panic)"; if (hooks && hooks->before_seams) hooks->before_seams(); return DMI_ERR_CONNECTIVITY; }
  out.init_rev.assign(w.init_corners.rbegin(), w.init_corners.rend());   // edgebreaker.rs:523-529
  pool_give(out.processed);
  out.processed.swap(w.processed);   // (the walker's own array: its remaining readers below go through `corner_at`)
  out.processed_quad = w.quad;
  out.seeds.clear();
  const std::vector<uint32_t>& processed = out.processed;
  const bool pq = w.quad;
  auto corner_at = [&processed, pq](size_t i) -> uint32_t { const uint32_t c = processed[i]; return pq ? c - (c >> 2) : c; };   // entry i of `processed` as a 3·face + k id
  if (hooks && hooks->seeds_ready) hooks->seeds_ready();
  s.leb128(w.symbols.size());
  s.leb128(w.num_split_symbols);
  {   // encode_topology_splits :375-403
    s.leb128(w.splits.size());
    uint64_t last = 0;
    for (auto& sp : w.splits) { s.leb128(sp.merging - last); s.leb128(sp.merging - sp.split); last = sp.merging; }
    BitPackerLsb bp(s.b);
    for (auto& sp : w.splits) bp.put(1, sp.right);
    bp.flush();
  }
  {   // DefaultTraversal::encode :575-656 — CLERS bits, reversed, LSB-first
    static const uint8_t len[5] = {1, 3, 3, 3, 3};
    static const uint8_t code[5] = {0, 0b1, 0b11, 0b101, 0b111};
    // (a 64-bit window written four bytes at a time into an array sized for three bits per symbol: a push_back per byte was 1 ns per face)
    const size_t n_sym = w.symbols.size();
    Pooled<uint8_t> bits_p((3 * n_sym + 7) / 8 + 16);
    std::vector<uint8_t>& bits = bits_p.v;
    if (bits.capacity() < (3 * n_sym + 7) / 8 + 16) bits.reserve((3 * n_sym + 7) / 8 + 16);
    uint8_t* const bp = bits.data();
    const uint8_t* const sym = w.symbols.data();
    uint64_t acc = 0;
    unsigned nb = 0;
    size_t at = 0;
    for (size_t i = n_sym; i-- > 0;) {
      const unsigned k = sym[i] & 7u;
      acc |= (uint64_t)code[k] << nb;
      nb += len[k];
      if (nb >= 32) { const uint32_t lo = (uint32_t)acc; std::memcpy(bp + at, &lo, 4); at += 4; acc >>= 32; nb -= 32; }
    }
    for (; nb > 0; nb = nb > 8 ? nb - 8 : 0) { bp[at++] = (uint8_t)acc; acc >>= 8; }
    set_size_written(bits, at);
    s.leb128(bits.size());
    s.bytes(bits);
  }
  // zero_prob, then the flags fed last to first (`fed` = the flags already in feeding order)
  auto rabs_block_fed = [&](const uint8_t* fed, size_t n, uint64_t zeros) -> bool {
    const uint8_t zp = zero_probability(zeros, (float)n);
    s.u8(zp);
    std::vector<uint8_t> bytes;
    if (!host_rabs_bytes(zp, fed, n, bytes)) return false;
    s.leb128(bytes.size());
    s.bytes(bytes);
    return true;
  };
  const double t_bits = since(t0);
  if (hooks && hooks->before_seams) hooks->before_seams();
  const double t_wait = since(t0);
  {
    std::vector<uint8_t> fed(w.start_interior.rbegin(), w.start_interior.rend());
    uint64_t zeros = 0;
    for (uint8_t b : fed) zeros += !b;
    if (!rabs_block_fed(fed.data(), fed.size(), zeros)) { err = "rABS state too large"; return DMI_ERR_ENTROPY; }
  }
  {   // attribute seams :611-653.  The reference walks the processed faces last to first, marks each visited and emits, for its three
      // corners in turn, the seam flag of every edge whose other face has not been visited yet.  "Not visited yet" only depends on the two
      // faces' positions in `processed`, so the flags are produced in parallel slices — straight into the order the rABS coder is fed in
      // (the reverse of the order of emission: faces first to last, corners prev, next, c) — and the streams of the attributes are
      // coded side by side on the multiply-high coder of host_chains.cpp (a divide per flag was 2/3 of this stage).
    const size_t n = processed.size(), A = t.att.size();
    // Which streams are distinct: an attribute without interior seams flags no emitted edge (they all have two faces) — ONE all-zero
    // stream serves every such attribute; an attribute copied from an earlier one repeats that one's stream.
    std::vector<int> stream_of(A, -1);   // attribute whose stream this one repeats (itself: its own flags), -1 = the all-zero stream
    std::vector<size_t> own;
    for (size_t j = 0; j < A; ++j) {
      if (!t.att[j].interior_seams) continue;
      const int a0 = t.att[j].alias_of;
      if (a0 >= 0 && (size_t)a0 < j && stream_of[(size_t)a0] == a0) stream_of[j] = a0;
      else { stream_of[j] = (int)j; own.push_back(j); }
    }
    uint64_t total = 0;                  // edges emitted (the length of every stream)
    const bool sliced = n >= (1u << 20) && A;          // (a small mesh — one of a batch, on its own thread — walks the loop as it stands)
    // every face is an interior start face or processed at least once: with exactly F of them no face was processed twice, and the masks the
    // traversal left in its symbols (bits 4–6: the edges towards faces processed earlier) are the edges the reference's walk from the back emits
    uint64_t interior_starts = 0;
    for (uint8_t b : w.start_interior) interior_starts += b;
    const bool masks_ok = n + interior_starts == t.F && w.symbols.size() == n && !std::getenv("DMI_NO_SEAM_MASKS");
    Pooled<uint32_t> where_p;                          // position of a face in `processed`
    std::vector<uint32_t>& where = where_p.v;
    std::atomic<int> twice{0};
    if (sliced && !masks_ok) {
      pool_fit(where, t.F);
      where.assign(t.F, kNone);
      parallel_for(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) where[corner_at(i) / 3] = (uint32_t)i; });
      parallel_for(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) if (where[corner_at(i) / 3] != (uint32_t)i) { twice.store(1); break; } });
    }
    std::vector<std::vector<uint8_t>> fed(A);
    struct GiveBack { std::vector<std::vector<uint8_t>>& v; ~GiveBack() { for (auto& x : v) pool_give(x); } } fed_back{fed};
    std::vector<uint64_t> zeros(A, 0);
    if (sliced && !twice.load() && own.empty()) {
      // no attribute has seams of its own: only the stream length is needed — every interior edge is emitted once (by the earlier of its two
      // faces in the walk from the back, or by the processed one when the other is a start face)
      if (t.no_boundary) total = (uint64_t)t.F * 3 / 2;
      else {
        std::atomic<uint64_t> inner{0};
        parallel_for((size_t)t.F * 3, [&](size_t lo, size_t hi) { uint64_t k = 0; for (size_t c = lo; c < hi; ++c) k += t.opp[c] != kNone; inner.fetch_add(k); });
        total = inner.load() / 2;
      }
    } else if (sliced && !twice.load()) {
      // corners of face i whose flag is emitted, as a mask over (c, next, prev)
      auto mask_of = [&](size_t i) -> uint32_t {
        if (masks_ok) return (uint32_t)(w.symbols[i] >> 4);
        const uint32_t c = corner_at(i);
        const uint32_t cs[3] = {c, corner_next(c), corner_prev(c)};
        uint32_t m = 0;
        for (int k = 0; k < 3; ++k) {
          const uint32_t o = w.opp3(cs[k]);
          if (o == kNone) continue;
          const uint32_t wo = where[o / 3];
          if (wo == kNone || wo < (uint32_t)i) m |= 1u << k;      // its face comes later in the walk from the back (or never: a start face)
        }
        return m;
      };
      constexpr size_t kChunk = 1u << 16;
      const size_t n_chunks = (n + kChunk - 1) / kChunk;
      Pooled<uint8_t> mask_p(n, (uint8_t)0);
      std::vector<uint8_t>& mask = mask_p.v;
      std::vector<uint64_t> chunk_sum(n_chunks + 1, 0);
      {
        std::vector<std::atomic<uint64_t>> acc(n_chunks);
        for (auto& a : acc) a.store(0);
        parallel_for(n, [&](size_t lo, size_t hi) {
          for (size_t i = lo; i < hi;) {
            const size_t end = std::min(hi, (i / kChunk + 1) * kChunk);
            uint64_t k = 0;
            for (size_t j = i; j < end; ++j) { mask[j] = (uint8_t)mask_of(j); k += (uint64_t)kBits3[mask[j] & 7u]; }
            acc[i / kChunk].fetch_add(k);
            i = end;
          }
        });
        for (size_t c = 0; c < n_chunks; ++c) chunk_sum[c + 1] = chunk_sum[c] + acc[c].load();
      }
      total = chunk_sum[n_chunks];
      for (size_t j : own) { pool_fit(fed[j], total); fed[j].resize(total); }
      std::vector<std::atomic<uint64_t>> zacc(A);
      for (auto& a : zacc) a.store(0);
      if (!own.empty()) parallel_for(n, [&](size_t lo, size_t hi) {
        uint64_t pos = chunk_sum[lo / kChunk];
        for (size_t j = (lo / kChunk) * kChunk; j < lo; ++j) pos += (uint64_t)kBits3[mask[j] & 7u];
        std::vector<uint64_t> z(A, 0);
        for (size_t i = lo; i < hi; ++i) {
          const uint32_t c = corner_at(i);
          const uint32_t cs[3] = {c, corner_next(c), corner_prev(c)};
          for (int k = 2; k >= 0; --k) {
            if (!(mask[i] >> k & 1u)) continue;
            for (size_t j : own) { const uint8_t f = t.att[j].seam_edge[cs[k]]; fed[j][pos] = f; z[j] += !f; }
            ++pos;
          }
        }
        for (size_t j = 0; j < A; ++j) zacc[j].fetch_add(z[j]);
      });
      for (size_t j = 0; j < A; ++j) zeros[j] = zacc[j].load();
    } else if (A && own.empty() && masks_ok) {   // no attribute has seams of its own: only the stream length — the edges the traversal recorded
      for (size_t i = 0; i < n; ++i) total += (uint64_t)kBits3[(w.symbols[i] >> 4) & 7u];
    } else if (A && own.empty() && [&] {   // a small mesh without seams: only the stream length — every face processed once ⇒ the interior-edge count
                 std::vector<uint8_t> seen(t.F, 0);
                 for (size_t i = 0; i < n; ++i) { uint8_t& f = seen[corner_at(i) / 3]; if (f) return false; f = 1; }
                 return true;
               }()) {
      if (t.no_boundary) total = (uint64_t)t.F * 3 / 2;
      else { uint64_t k = 0; for (size_t c = 0; c < (size_t)t.F * 3; ++c) k += t.opp[c] != kNone; total = k / 2; }
    } else if (A) {   // small meshes with seams, and a face processed twice (malformed tables): the reference's loop as it stands
      // (the flags are written from the back of arrays sized for every corner — the coder is fed in the reverse of the order of emission — and
      //  moved to the front at the end: no growing vectors, no reversing copy)
      const size_t cap = 3 * n;
      for (size_t j : own) { pool_fit(fed[j], cap); fed[j].resize(cap); }
      size_t at = cap;
      if (masks_ok) {   // no table look-ups: the traversal recorded which edges every face emits
        for (size_t i = n; i-- > 0;) {
          const uint32_t m = w.symbols[i] >> 4;
          if (!m) continue;
          const uint32_t c = corner_at(i);
          const uint32_t cs[3] = {c, corner_next(c), corner_prev(c)};
          for (int k = 0; k < 3; ++k) {
            if (!(m >> k & 1u)) continue;
            ++total;
            --at;
            for (size_t j : own) { const uint8_t f = t.att[j].seam_edge[cs[k]]; fed[j][at] = f; zeros[j] += !f; }
          }
        }
      } else {
        std::vector<uint8_t> fv(t.F, 0);
        for (size_t i = n; i-- > 0;) {
          const uint32_t c = corner_at(i);
          const uint32_t cs[3] = {c, corner_next(c), corner_prev(c)};
          fv[c / 3] = 1;
          for (uint32_t cc : cs) {
            const uint32_t o = w.opp3(cc);
            if (o == kNone || fv[o / 3]) continue;
            ++total;
            --at;
            for (size_t j : own) { const uint8_t f = t.att[j].seam_edge[cc]; fed[j][at] = f; zeros[j] += !f; }
          }
        }
      }
      for (size_t j : own) { std::memmove(fed[j].data(), fed[j].data() + at, (size_t)total); fed[j].resize((size_t)total); }
    }
    // (the stream of an attribute without seams is `total` zero flags: coded by its period, host_rabs_constant — no flag array, no 1.5 steps per face)
    const bool need_zero = own.size() < A && std::find(stream_of.begin(), stream_of.end(), -1) != stream_of.end();
    // one coder per attribute, side by side for large meshes
    std::vector<std::vector<uint8_t>> coded(A + 1);   // [A] = the all-zero stream
    std::vector<uint8_t> ok(A + 1, 1), zp(A + 1, 0);
    auto code_one = [&](size_t j) {
      zp[j] = zero_probability(j == A ? total : zeros[j], (float)total);
      ok[j] = (j == A ? host_rabs_constant(zp[j], 0u, total, coded[j]) : host_rabs_bytes(zp[j], fed[j].data(), total, coded[j])) ? 1 : 0;
    };
    std::vector<size_t> todo(own);
    if (need_zero) todo.push_back(A);
    if (todo.size() > 1 && n >= (1u << 20) && host_threads() > 1) {
      std::vector<std::thread> th;
      for (size_t j : todo) th.emplace_back(code_one, j);
      for (auto& x : th) x.join();
    } else for (size_t j : todo) code_one(j);
    for (size_t j = 0; j < A; ++j) {
      const size_t from = stream_of[j] < 0 ? A : (size_t)stream_of[j];
      if (!ok[from]) { err = "rABS state too large"; return DMI_ERR_ENTROPY; }
      s.u8(zp[from]);
      s.leb128(coded[from].size());
      s.bytes(coded[from]);
    }
  }
  if (trace_all) { g_eb_ns[0] += (uint64_t)(t_ctor * 1e6); g_eb_ns[1] += (uint64_t)((t_walk - t_ctor) * 1e6); g_eb_ns[2] += (uint64_t)((t_bits - t_walk) * 1e6); g_eb_ns[3] += (uint64_t)((since(t0) - t_wait) * 1e6); g_eb_ns[4] += 1; }
  if (trace) std::fprintf(stderr, "[dmi]   Edgebreaker of %u faces: boundaries + traversal %.1f ms, seeds + CLERS bits %.1f, wait for the seam flags %.1f, seam streams %.1f\n", t.F, t_walk,
                          t_bits - t_walk, t_wait - t_bits, since(t0) - t_wait);
  out.connectivity.swap(s.b);
  return DMI_OK;
}

// Attribute sequencer (shared/attribute/sequence.rs:48-151) as a plain O(F) depth-first walk.
// The reference also deletes every stack entry lying in the face it has just marked visited;
// such entries are skipped on pop anyway, so the emitted order is unchanged without the deletion.
void vertex_boundary_flags(const TableRef& t, std::vector<uint8_t>& on_boundary) {
  pool_fit(on_boundary, t.V);
  on_boundary.assign(t.V, 0);
  parallel_for(t.V, [&](size_t lo, size_t hi) {
    for (size_t v = lo; v < hi; ++v) { const uint32_t l0 = t.lmc[v]; on_boundary[v] = (l0 != kNone && t.opp[corner_next(l0)] == kNone) ? 1 : 0; }
  });
}

void attribute_sequence(const TableRef& t, const uint32_t* seeds, uint32_t n_seeds, std::vector<uint32_t>& seq, const uint8_t* on_boundary) {
  attribute_sequence(t, nullptr, 0, seeds, n_seeds, seq, on_boundary, false);
}
// The reference's stack starts as the seeds and only ever grows above them: what the walk pushes is popped before the next seed.  The seeds
// are therefore read in place, last to first (second part, then first part), and only the pushes live on a stack of their own.
// kStamp (tables of ≥ 2^16 faces): the per-vertex state is a 32-bit stamp — position in `seq` + 1, bit 31 = the vertex lies on a boundary — instead of a
// flag byte: a step whose tip was emitted by the spiral's previous loop then knows where in `seq` that loop passed, a shadow index follows the walk one
// loop behind, and the table lines of the corner that loop emitted a few entries later are requested into L1 (the traversal's trick, Walker::run_from_t:
// 10M faces 44.5 → ≈ 40.5 ms).
// E: the form of the corner ids `t.opp` holds (Enc3 / Enc4, see Walker); second_quad: the second part of the seeds holds 4·face + k ids (a traversal over
// such a table leaves them so).  What the walk EMITS are 3·face + k ids in either form.
// A face's single successor is walked next without a trip through the stack (the reference pushes it and pops it at once: a store, a load and the
// forwarding between them on the walk's dependency chain), and the walk is over when every vertex has been emitted.  (Tried: skipping the reference's test
// of a corner's next / previous vertices for corners that come off the walk's own stack — they lie across an edge of a face just processed, both vertices
// visited — for tables the device checked: no change, the step is bound by the load of the opposite corner.)
template <bool kStamp, class E, bool kClosed>
static void sequence_impl(const TableRef& t, const uint32_t* first, uint32_t n_first, const uint32_t* second, uint32_t n_second, bool second_quad, std::vector<uint32_t>& seq, const uint8_t* on_boundary) {
  Pooled<uint8_t> vvis_p(kStamp ? 0 : t.V, (uint8_t)0), fvis_p(t.F, (uint8_t)0);
  Pooled<uint32_t> vst_p(kStamp ? t.V : 0, 0u);
  constexpr uint32_t kOnBoundary = 0x80000000u, kPos = 0x7FFFFFFFu;
  // bytes: bit 0 visited, bit 1 the vertex lies on a boundary (from on_boundary: one load less per new vertex)
  if (on_boundary && !kClosed) {
    if (kStamp) parallel_for(t.V, [&](size_t lo, size_t hi) { for (size_t v = lo; v < hi; ++v) vst_p.v[v] = on_boundary[v] ? kOnBoundary : 0u; });
    else parallel_for(t.V, [&](size_t lo, size_t hi) { for (size_t v = lo; v < hi; ++v) vvis_p.v[v] = on_boundary[v] ? 2 : 0; });
  }
  // (tables, flags, the output and the stack in locals, written through raw pointers: see Walker::run_from)
  const uint32_t* const opp = t.opp;
  const uint32_t* const c2v = t.c2v;
  uint8_t* const vv = vvis_p.v.data();
  uint32_t* const vs = vst_p.v.data();
  uint8_t* const fv = fvis_p.v.data();
  pool_fit(seq, t.V);
  if (seq.capacity() < (size_t)t.V) seq.reserve(t.V);
  uint32_t* const sq = seq.data();
  size_t nq = 0;
  const size_t qcap = t.V;             // every vertex is emitted once (sequence.rs:41-46: its flag is set with the emission), so nq never passes t.V
  std::vector<uint32_t> stack_store(4096);
  uint32_t* st = stack_store.data();
  size_t sn = 0, scap = stack_store.size();
  auto push = [&](uint32_t x) {
    if (__builtin_expect(sn == scap, 0)) { stack_store.resize(scap * 2); st = stack_store.data(); scap *= 2; }
    st[sn++] = x;
  };
  uint64_t left = (uint64_t)n_first + n_second;
  size_t q = ~(size_t)0 >> 1;          // the shadow: position in `seq` of what the previous loop emitted beside this spot
  constexpr size_t kAhead = 8;
  auto visited = [&](uint32_t v) -> bool { return kStamp ? (vs[v] & kPos) != 0u : (vv[v] & 1) != 0; };
  auto emit = [&](uint32_t i) {   // i: where the corner's entries are (= its 3·face + k id)
    const uint32_t v = c2v[i];
    if (kStamp) { const uint32_t f = vs[v]; if (!(f & kPos)) { sq[nq] = i; vs[v] = f | (uint32_t)++nq; } }
    else { const uint8_t f = vv[v]; if (!(f & 1)) { vv[v] = f | 1; sq[nq++] = i; } }
  };
  uint32_t c;
  for (;;) {
    if (sn) c = st[--sn];
    else {
      // Seeds.  Once every vertex is emitted nothing can follow — the walk from the traversal's last corner covers a whole component before the next seed
      // is read: a one-component mesh would otherwise read F more seeds only to skip them (10 ms per 10M faces).
      if (nq == qcap || !left) break;
      --left;
      if (left >= n_first) { c = second[left - n_first]; if (second_quad != E::kQuad) c = second_quad ? E::from3(Enc4::to3(c)) : E::from3(c); }
      else c = E::from3(first[left]);
    }
  have_c:
    const uint32_t f = E::face(c);
    if (fv[f]) continue;
    const uint32_t k = E::k(c, f), i = E::idx(c, f);
    prefetch_neighbours(opp + i); prefetch_neighbours(c2v + i);
    const uint32_t nc = k == 2 ? i - 2 : i + 1, pc = k == 0 ? i + 2 : i - 1;   // (the next / previous corner's entries: 3·face + k ids)
    if (!visited(c2v[nc]) || !visited(c2v[pc])) { emit(nc); emit(pc); goto have_c; }   // (pushed and popped at once by the reference)
    if (kStamp) {
      const size_t qa = q + kAhead;
      if (qa < nq) { const uint32_t g = sq[qa]; __builtin_prefetch(opp + g, 0, 3); __builtin_prefetch(opp + g + 16, 0, 3); __builtin_prefetch(opp + g - 16, 0, 3); __builtin_prefetch(c2v + g, 0, 3); }
    }
    fv[f] = 1;
    const uint32_t v = c2v[i];
    const uint32_t right = opp[nc], lft = opp[pc];
    const uint32_t vflags = kStamp ? vs[v] : (uint32_t)vv[v];
    if (kStamp ? !(vflags & kPos) : !(vflags & 1)) {
      if (kStamp) { sq[nq] = i; vs[v] = vflags | (uint32_t)++nq; } else { vv[v] = (uint8_t)(vflags | 1); sq[nq++] = i; }
      ++q;
      bool boundary;
      if (kClosed) boundary = false;
      else if (on_boundary) boundary = kStamp ? (vflags & kOnBoundary) != 0u : (vflags & 2) != 0;
      else { const uint32_t l0 = t.lmc[v]; boundary = opp[corner_next(l0)] == kNone; }   // is_on_boundary: swing_left(lmc) is None
      if (!boundary) { if (kClosed || right != kNone) { c = right; goto have_c; } continue; }
    } else if (kStamp) {
      q = (size_t)(vflags & kPos);   // (the tip was emitted at position (vflags & kPos) - 1: the shadow moves on from the entry after it)
    }
    const bool r_has = kClosed || right != kNone, l_has = kClosed || lft != kNone;
    const bool rdone = r_has && fv[E::face(right)], ldone = l_has && fv[E::face(lft)];
    if (rdone) { if (!ldone && l_has) { c = lft; goto have_c; } }
    else if (ldone) { if (r_has) { c = right; goto have_c; } }
    else if (l_has && r_has) { push(lft); c = right; goto have_c; }
    else if (l_has) { c = lft; goto have_c; }
    else if (r_has) { c = right; goto have_c; }
  }
  set_size_written(seq, nq);
}
void attribute_sequence(const TableRef& t, const uint32_t* first, uint32_t n_first, const uint32_t* second, uint32_t n_second, std::vector<uint32_t>& seq, const uint8_t* on_boundary,
                        bool second_quad) {
  static const bool no_shadow = std::getenv("DMI_NO_SHADOW") != nullptr || std::getenv("DMI_NO_SEQ_SHADOW") != nullptr;
  const bool stamps = t.F >= (1u << 16) && t.V < 0x7FFFFFF0u && !no_shadow;
  // (closed: every corner has an opposite — the walk then tests no entry for "none" and no vertex for "on a boundary")
#define DMI_SEQ(S, E, C) sequence_impl<S, E, C>(t, first, n_first, second, n_second, second_quad, seq, on_boundary)
  static const bool no_closed = std::getenv("DMI_NO_CLOSED") != nullptr;
  const bool closed = t.closed && !no_closed;
  if (t.quad) { if (stamps) { if (closed) DMI_SEQ(true, Enc4, true); else DMI_SEQ(true, Enc4, false); } else DMI_SEQ(false, Enc4, false); }
  else { if (stamps) { if (closed) DMI_SEQ(true, Enc3, true); else DMI_SEQ(true, Enc3, false); } else DMI_SEQ(false, Enc3, false); }
#undef DMI_SEQ
}

}  // namespace dmi
