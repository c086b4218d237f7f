// host_conn.cpp — the two serial walks of the connectivity stage: Edgebreaker traversal + connectivity bytes (encode/connectivity/edgebreaker.rs), attribute
// sequencer (shared/attribute/sequence.rs).  CPU, one thread per walk, flat arrays (the tables: host_tables.cpp or the device stage, dmi_conn.hip).  Output is
// bit-identical to the reference's (quirks kept: per-vertex "hole" ids).
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
    const bool no_shadow = dbg_on(DMI_DBG_NO_SHADOW);
    const uint32_t min_faces = dbg().shadow_min_faces ? dbg().shadow_min_faces : (1u << 16);   // (256-mesh batch: traversal thread time 71 → 68 ms with the meshes of ≥ 2^16 faces on stamps; below, tables and flags sit in L2)
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
    const bool no_closed = dbg_on(DMI_DBG_NO_CLOSED);
    const bool closed = t.no_boundary && !no_closed;
    if (quad) {
      if (use_stamp) { if (closed) run_from_t<true, Enc4, true>(Enc4::from3(c3)); else run_from_t<true, Enc4, false>(Enc4::from3(c3)); }
      else { if (closed) run_from_t<false, Enc4, true>(Enc4::from3(c3)); else run_from_t<false, Enc4, false>(Enc4::from3(c3)); }
    } else {
      if (use_stamp) { if (closed) run_from_t<true, Enc3, true>(c3); else run_from_t<true, Enc3, false>(c3); }
      else { if (closed) run_from_t<false, Enc3, true>(c3); else run_from_t<false, Enc3, false>(c3); }
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
          const uint8_t rf = r_none ? 1 : fv[E::face(rc)], lf = l_none ? 1 : fv[E::face(lc)];
          rv = rf & 1; lv = lf & 1;
          r_split = !r_none && (rf & 2); l_split = !l_none && (lf & 2);
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
  const bool trace_all = dbg_on(DMI_DBG_TRACE);
  struct Whole { bool on; std::chrono::steady_clock::time_point a = std::chrono::steady_clock::now(); ~Whole() { if (on) g_eb_ns[5] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - a).count(); } } whole{trace_all};   // (declared first: its destructor runs after every other local's)
  const bool trace = t.F > 100000 && dbg_on(DMI_DBG_TRACE);
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
    const bool masks_ok = n + interior_starts == t.F && w.symbols.size() == n && !dbg_on(DMI_DBG_NO_SEAM_MASKS);
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
        // Seams are rare (a fraction of a percent of the edges): a byte per FACE — does any of its three edges lie on a seam of any of these attributes? —
        // lets every other face pass with one sequential look-up instead of a gather and a store per emitted edge (≈ 8 of this step's ≈ 14 ns per face;
        // the flag arrays start as zeros, only the faces on a seam write into them)
        Pooled<uint8_t> fsum_p(t.F, (uint8_t)0);
        uint8_t* const fsum = fsum_p.v.data();
        for (size_t j : own) {
          const uint8_t* se = t.att[j].seam_edge.data();
          for (size_t f = 0; f < t.F; ++f) fsum[f] |= (uint8_t)(se[3 * f] | se[3 * f + 1] | se[3 * f + 2]);
          std::memset(fed[j].data(), 0, cap);
        }
        for (size_t i = n; i-- > 0;) {
          const uint32_t m = w.symbols[i] >> 4;
          if (!m) continue;
          const uint32_t c = corner_at(i);
          if (!fsum[c / 3]) { const uint32_t k = kBits3[m & 7u]; total += k; at -= k; for (size_t j : own) zeros[j] += k; continue; }
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
      std::vector<dmi::Thread> th;
      for (size_t j : todo) th.emplace_back(with_debug(code_one), j);
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
  attribute_sequence(t, nullptr, 0, seeds, n_seeds, seq, on_boundary, false, nullptr);
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
static void sequence_impl(const TableRef& t, const uint32_t* first, uint32_t n_first, const uint32_t* second, uint32_t n_second, bool second_quad, std::vector<uint32_t>& seq, const uint8_t* on_boundary,
                          SeqProgress* progress) {
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
  if (progress) progress->host.store(sq, std::memory_order_release);
  // (published every 2^16 entries: a well-predicted branch on a register beside ≈ 7 ns of table look-ups per entry)
  auto publish = [&](size_t n) { if (__builtin_expect(progress != nullptr, 0) && !(n & 0xFFFFu)) progress->written.store((uint32_t)n, std::memory_order_release); };
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
    if (kStamp) { const uint32_t f = vs[v]; if (!(f & kPos)) { sq[nq] = i; vs[v] = f | (uint32_t)++nq; publish(nq); } }
    else { const uint8_t f = vv[v]; if (!(f & 1)) { vv[v] = f | 1; sq[nq++] = i; publish(nq); } }
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
      publish(nq);
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
                        bool second_quad, SeqProgress* progress) {
  const bool no_shadow = dbg_on(DMI_DBG_NO_SHADOW | DMI_DBG_NO_SEQ_SHADOW);
  const bool stamps = t.F >= (1u << 16) && t.V < 0x7FFFFFF0u && !no_shadow;
  // (closed: every corner has an opposite — the walk then tests no entry for "none" and no vertex for "on a boundary")
#define DMI_SEQ(S, E, C) sequence_impl<S, E, C>(t, first, n_first, second, n_second, second_quad, seq, on_boundary, progress)
  const bool no_closed = dbg_on(DMI_DBG_NO_CLOSED);
  const bool closed = t.closed && !no_closed;
  if (t.quad) {
    if (stamps) { if (closed) DMI_SEQ(true, Enc4, true); else DMI_SEQ(true, Enc4, false); }
    else { if (closed) DMI_SEQ(false, Enc4, true); else DMI_SEQ(false, Enc4, false); }
  } else {
    if (stamps) { if (closed) DMI_SEQ(true, Enc3, true); else DMI_SEQ(true, Enc3, false); }
    else { if (closed) DMI_SEQ(false, Enc3, true); else DMI_SEQ(false, Enc3, false); }
  }
#undef DMI_SEQ
}

}  // namespace dmi
