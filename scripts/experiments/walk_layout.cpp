// walk_layout.cpp — layout experiment for the two serial host walks (Edgebreaker traversal, attribute sequencer) on a closed
// torus grid: separate opp / c2v / flag arrays (the production layout of host_conn.cpp) against ONE 32-byte record per face
// (corner ids 8·face + k, the face's flags inside its record).  Same visiting order in both forms (checked).  CPU only.
//   g++ -O2 -std=c++17 -o walk_layout walk_layout.cpp && ./walk_layout [n=2236] [repeats=5]
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/mman.h>

static constexpr uint32_t kNone = 0xFFFFFFFFu;
static inline uint32_t cnext(uint32_t c) { return (c % 3 == 2) ? c - 2 : c + 1; }
static inline uint32_t cprev(uint32_t c) { return (c % 3 == 0) ? c + 2 : c - 1; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class T> static T* huge_alloc(size_t n) {
  const size_t bytes = ((n * sizeof(T) + (2u << 20) - 1) >> 21) << 21;
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { std::perror("mmap"); std::exit(1); }
  madvise(p, bytes, MADV_HUGEPAGE);
  std::memset(p, 0, bytes);
  return static_cast<T*>(p);
}

static int kPf = 16;
static inline void pf(const uint32_t* p) { if (kPf) { __builtin_prefetch(p + kPf, 0, 2); __builtin_prefetch(p - kPf, 0, 2); } }
static inline void pfb(const uint8_t* p) { if (kPf) { __builtin_prefetch(p + 4 * kPf, 1, 2); __builtin_prefetch(p - 4 * kPf, 1, 2); } }

enum : uint8_t { SYM_C, SYM_S, SYM_L, SYM_R, SYM_E };

// ---------------- production layout ----------------
struct Flat {
  uint32_t F, V;
  uint32_t *opp, *c2v;
  uint8_t *fvis, *vvis;
  uint32_t* processed; uint8_t* symbols; size_t n_processed = 0;
  std::vector<uint32_t> stack;
  void run_from(uint32_t c) {
    stack.clear(); stack.push_back(c);
    while (!stack.empty()) {
      c = stack.back();
      if (fvis[c / 3] & 1) { stack.pop_back(); continue; }
      for (;;) {
        pf(opp + c); pf(c2v + c);
        const uint32_t f = c / 3, v = c2v[c];
        pfb(fvis + f); pfb(vvis + v);
        fvis[f] |= 1;
        processed[n_processed] = c;
        const uint32_t gate = opp[c] != kNone ? 0x10u : 0u;
        const uint8_t vflags = vvis[v];
        if (!(vflags & 1)) {
          vvis[v] = vflags | 1;
          if (!(vflags & 2)) { symbols[n_processed++] = (uint8_t)(SYM_C | gate); c = opp[cnext(c)]; continue; }
        }
        const uint32_t rc = opp[cnext(c)], lc = opp[cprev(c)];
        const bool rv = rc == kNone || (fvis[rc / 3] & 1), lv = lc == kNone || (fvis[lc / 3] & 1);
        const uint8_t nb = (uint8_t)(gate | ((rc != kNone && rv) ? 0x20u : 0u) | ((lc != kNone && lv) ? 0x40u : 0u));
        if (rv) {
          if (lv) { symbols[n_processed++] = (uint8_t)(SYM_E | nb); stack.pop_back(); break; }
          symbols[n_processed++] = (uint8_t)(SYM_R | nb); c = lc;
        } else if (lv) { symbols[n_processed++] = (uint8_t)(SYM_L | nb); c = rc; }
        else { symbols[n_processed++] = (uint8_t)(SYM_S | nb); fvis[f] |= 2; stack.back() = lc; stack.push_back(rc); break; }
      }
    }
  }
  void edgebreaker() {
    n_processed = 0;
    for (uint32_t f = 0; f < F; ++f) {
      if (fvis[f] & 1) continue;
      const uint32_t start = 3 * f;   // closed mesh: interior start
      vvis[c2v[start]] |= 1; vvis[c2v[start + 1]] |= 1; vvis[c2v[start + 2]] |= 1;
      fvis[f] |= 1;
      run_from(opp[cnext(start)]);
    }
  }
  // sequencer: flags in bit 2 (0x4) of the same byte arrays
  size_t sequence(uint32_t* seq) {
    size_t n_seq = 0;
    uint64_t left = n_processed;
    stack.clear();
    auto emit = [&](uint32_t c) { const uint32_t v = c2v[c]; if (!(vvis[v] & 4)) { vvis[v] |= 4; seq[n_seq++] = c; } };
    for (;;) {
      uint32_t c;
      if (!stack.empty()) { c = stack.back(); stack.pop_back(); }
      else if (left) { --left; c = processed[left]; }
      else break;
      if (fvis[c / 3] & 4) continue;
      pf(opp + c); pf(c2v + c); pfb(fvis + c / 3);
      const uint32_t nc = cnext(c), pc = cprev(c);
      if (!(vvis[c2v[nc]] & 4) || !(vvis[c2v[pc]] & 4)) { emit(nc); emit(pc); stack.push_back(c); continue; }
      fvis[c / 3] |= 4;
      const uint32_t v = c2v[c];
      pfb(vvis + v);
      const uint32_t right = opp[nc], lft = opp[pc];
      const uint8_t vflags = vvis[v];
      if (!(vflags & 4)) {
        emit(c);
        if (!(vflags & 2)) { if (right != kNone) stack.push_back(right); continue; }
      }
      const bool rdone = right != kNone && (fvis[right / 3] & 4), ldone = lft != kNone && (fvis[lft / 3] & 4);
      if (rdone) { if (!ldone && lft != kNone) stack.push_back(lft); }
      else if (ldone) { if (right != kNone) stack.push_back(right); }
      else { if (lft != kNone) stack.push_back(lft); if (right != kNone) stack.push_back(right); }
    }
    return n_seq;
  }
};


// ---------------- production-like loop (host_conn.cpp's Walker::run_from as it stands): push_back outputs, guards, split bookkeeping ----------------
#include <unordered_map>
template <bool PUSH, bool GUARD, bool SPLIT>
struct Prod {
  uint32_t F, V;
  const uint32_t *opp, *c2v;
  std::vector<uint8_t> vvis, fvis;
  std::vector<uint32_t> stack, processed;
  std::vector<uint8_t> symbols;
  uint32_t* praw; uint8_t* sraw; size_t n_out = 0;
  std::unordered_map<uint32_t, uint64_t> split_symbol_of_face;
  struct Split { uint64_t merging, split; uint8_t right; };
  std::vector<Split> splits;
  uint64_t symbol_idx = ~0ull, num_split_symbols = 0;
  bool bad = false;
  void note_split(uint64_t merging, uint8_t right, uint32_t face) { if (SPLIT) if (fvis[face] & 2) splits.push_back({merging, split_symbol_of_face[face], right}); }
  inline void out(uint32_t c, uint8_t s) { if (PUSH) { processed.push_back(c); symbols.push_back(s); } else { praw[n_out] = c; sraw[n_out++] = s; } }
  void run_from(uint32_t c) {
    stack.clear(); stack.push_back(c);
    while (!stack.empty() && !bad) {
      c = stack.back();
      if (GUARD && c == kNone) { bad = true; return; }
      if (fvis[c / 3] & 1) { stack.pop_back(); continue; }
      for (uint32_t steps = 0; !GUARD || steps < F; ++steps) {
        if (GUARD && c == kNone) { bad = true; return; }
        ++symbol_idx;
        pf(opp + c); pf(c2v + c);
        const uint32_t f = c / 3, v = c2v[c];
        pfb(fvis.data() + f); pfb(vvis.data() + v);
        fvis[f] |= 1;
        const uint32_t gate = opp[c] != kNone ? 0x10u : 0u;
        const uint8_t vflags = vvis[v];
        if (!(vflags & 1)) {
          vvis[v] = vflags | 1;
          if (!(vflags & 2)) { out(c, (uint8_t)(SYM_C | gate)); c = opp[cnext(c)]; continue; }
        }
        const uint32_t rc = opp[cnext(c)], lc = opp[cprev(c)];
        const bool rv = rc == kNone || (fvis[rc / 3] & 1), lv = lc == kNone || (fvis[lc / 3] & 1);
        const uint8_t nb = (uint8_t)(gate | ((rc != kNone && rv) ? 0x20u : 0u) | ((lc != kNone && lv) ? 0x40u : 0u));
        if (rv) {
          if (rc != kNone) note_split(symbol_idx, 1, rc / 3);
          if (lv) {
            if (lc != kNone) note_split(symbol_idx, 0, lc / 3);
            out(c, (uint8_t)(SYM_E | nb)); stack.pop_back(); break;
          }
          out(c, (uint8_t)(SYM_R | nb)); c = lc;
        } else if (lv) {
          if (lc != kNone) note_split(symbol_idx, 0, lc / 3);
          out(c, (uint8_t)(SYM_L | nb)); c = rc;
        } else {
          out(c, (uint8_t)(SYM_S | nb)); ++num_split_symbols;
          if (SPLIT) split_symbol_of_face[f] = symbol_idx;
          fvis[f] |= 2; stack.back() = lc; stack.push_back(rc); break;
        }
      }
    }
  }
  double edgebreaker(uint32_t* pr, uint8_t* sr) {
    praw = pr; sraw = sr; n_out = 0;
    vvis.assign(V, 0); fvis.assign(F, 0);
    processed.clear(); symbols.clear(); processed.reserve(F); symbols.reserve(F);
    const double t0 = now_ms();
    for (uint32_t f = 0; f < F && !bad; ++f) {
      if (fvis[f] & 1) continue;
      const uint32_t start = 3 * f;
      vvis[c2v[start]] |= 1; vvis[c2v[start + 1]] |= 1; vvis[c2v[start + 2]] |= 1;
      fvis[f] |= 1;
      run_from(opp[cnext(start)]);
    }
    return now_ms() - t0;
  }
};
template <bool PUSH, bool GUARD, bool SPLIT>
static void run_prod(const char* name, uint32_t F, uint32_t V, const uint32_t* opp, const uint32_t* c2v, uint32_t* pr, uint8_t* sr, int repeats) {
  double best = 1e30;
  for (int r = 0; r < repeats; ++r) { Prod<PUSH, GUARD, SPLIT> p{F, V, opp, c2v}; best = std::min(best, p.edgebreaker(pr, sr)); }
  std::printf("  production-like Edgebreaker (%s): %.1f ms (%.2f ns/face)\n", name, best, best * 1e6 / F);
}


// ---------------- right / left tables: RL[2c] = opp[next(c)], RL[2c + 1] = opp[prev(c)] (the hop of a step is ONE load at an address the previous hop
// delivered; the gate flag — opp[c] != none — rides in bit 31 of cv[c]) ----------------
struct HopTables {
  uint32_t F, V;
  const uint32_t *RL, *cv;
  uint8_t *fvis, *vvis;
  uint32_t* processed; uint8_t* symbols; size_t n_processed = 0;
  std::vector<uint32_t> stack;
  void run_from(uint32_t c) {
    stack.clear(); stack.push_back(c);
    while (!stack.empty()) {
      c = stack.back();
      if (fvis[c / 3] & 1) { stack.pop_back(); continue; }
      for (;;) {
        pf(RL + 2 * (size_t)c); pf(RL + 2 * (size_t)c + 16); pf(cv + c);
        const uint32_t f = c / 3, vg = cv[c], v = vg & 0x7FFFFFFFu;
        pfb(fvis + f); pfb(vvis + v);
        fvis[f] |= 1;
        processed[n_processed] = c;
        const uint32_t gate = (vg >> 31) << 4;
        const uint8_t vflags = vvis[v];
        const uint32_t rc = RL[2 * (size_t)c];
        if (!(vflags & 1)) {
          vvis[v] = vflags | 1;
          if (!(vflags & 2)) { symbols[n_processed++] = (uint8_t)(SYM_C | gate); c = rc; continue; }
        }
        const uint32_t lc = RL[2 * (size_t)c + 1];
        const bool rv = rc == kNone || (fvis[rc / 3] & 1), lv = lc == kNone || (fvis[lc / 3] & 1);
        const uint8_t nb = (uint8_t)(gate | ((rc != kNone && rv) ? 0x20u : 0u) | ((lc != kNone && lv) ? 0x40u : 0u));
        if (rv) {
          if (lv) { symbols[n_processed++] = (uint8_t)(SYM_E | nb); stack.pop_back(); break; }
          symbols[n_processed++] = (uint8_t)(SYM_R | nb); c = lc;
        } else if (lv) { symbols[n_processed++] = (uint8_t)(SYM_L | nb); c = rc; }
        else { symbols[n_processed++] = (uint8_t)(SYM_S | nb); fvis[f] |= 2; stack.back() = lc; stack.push_back(rc); break; }
      }
    }
  }
  void edgebreaker() {
    n_processed = 0;
    for (uint32_t f = 0; f < F; ++f) {
      if (fvis[f] & 1) continue;
      const uint32_t start = 3 * f;
      vvis[cv[start] & 0x7FFFFFFFu] |= 1; vvis[cv[start + 1] & 0x7FFFFFFFu] |= 1; vvis[cv[start + 2] & 0x7FFFFFFFu] |= 1;
      fvis[f] |= 1;
      run_from(RL[2 * (size_t)start]);
    }
  }
  size_t sequence(uint32_t* seq) {
    size_t n_seq = 0;
    uint64_t left = n_processed;
    stack.clear();
    auto emit = [&](uint32_t c) { const uint32_t v = cv[c] & 0x7FFFFFFFu; if (!(vvis[v] & 4)) { vvis[v] |= 4; seq[n_seq++] = c; } };
    for (;;) {
      uint32_t c;
      if (!stack.empty()) { c = stack.back(); stack.pop_back(); }
      else if (left) { --left; c = processed[left]; }
      else break;
      if (fvis[c / 3] & 4) continue;
      pf(RL + 2 * (size_t)c); pf(RL + 2 * (size_t)c + 16); pf(cv + c); pfb(fvis + c / 3);
      const uint32_t nc = cnext(c), pc = cprev(c);
      if (!(vvis[cv[nc] & 0x7FFFFFFFu] & 4) || !(vvis[cv[pc] & 0x7FFFFFFFu] & 4)) { emit(nc); emit(pc); stack.push_back(c); continue; }
      fvis[c / 3] |= 4;
      const uint32_t v = cv[c] & 0x7FFFFFFFu;
      pfb(vvis + v);
      const uint32_t right = RL[2 * (size_t)c], lft = RL[2 * (size_t)c + 1];
      const uint8_t vflags = vvis[v];
      if (!(vflags & 4)) {
        emit(c);
        if (!(vflags & 2)) { if (right != kNone) stack.push_back(right); continue; }
      }
      const bool rdone = right != kNone && (fvis[right / 3] & 4), ldone = lft != kNone && (fvis[lft / 3] & 4);
      if (rdone) { if (!ldone && lft != kNone) stack.push_back(lft); }
      else if (ldone) { if (right != kNone) stack.push_back(right); }
      else { if (lft != kNone) stack.push_back(lft); if (right != kNone) stack.push_back(right); }
    }
    return n_seq;
  }
};


// ---------------- the library's lean loop (host_conn.cpp Walker::run_from, round 4b) verbatim ----------------
struct LibWalker {
  uint32_t F, V;
  const uint32_t *t_opp, *t_c2v;
  std::vector<uint8_t> vvis, fvis, hole_done;
  std::vector<uint32_t> hole_of, stack, processed;
  std::vector<uint8_t> symbols;
  std::unordered_map<uint32_t, uint64_t> split_symbol_of_face;
  struct Split { uint64_t merging, split; uint8_t right; };
  std::vector<Split> splits;
  uint64_t num_split_symbols = 0;
  size_t n_out = 0;
  bool bad = false;
  __attribute__((noinline)) void note_split(uint64_t merging, uint8_t right, uint32_t face) { splits.push_back({merging, split_symbol_of_face[face], right}); }
  __attribute__((noinline)) void split_here(uint32_t c, uint32_t f, uint32_t v, uint8_t vflags, uint32_t rc, uint32_t lc, uint64_t symbol_idx) {
    ++num_split_symbols;
    if ((vflags & 2) && !hole_done[hole_of[v]]) bad = true;
    split_symbol_of_face[f] = symbol_idx;
    fvis[f] |= 2;
    stack.back() = lc;
    stack.push_back(rc);
  }
  __attribute__((noinline)) void run_from(uint32_t c) {
    const uint32_t* const opp = t_opp;
    const uint32_t* const c2v = t_c2v;
    uint8_t* const fv = fvis.data();
    uint8_t* const vv = vvis.data();
    uint32_t* const proc = processed.data();
    uint8_t* const sym = symbols.data();
    const size_t cap = F;
    size_t n = n_out;
    stack.clear();
    stack.push_back(c);
    while (!stack.empty() && !bad) {
      c = stack.back();
      if (c == kNone) { bad = true; break; }
      if (fv[c / 3] & 1) { stack.pop_back(); continue; }
      for (;;) {
#ifndef LIB_NO_GUARD
        if (c == kNone || n >= cap) { bad = true; break; }
#endif
        pf(opp + c); pf(c2v + c);
        const uint32_t f = c / 3, k = c - 3 * f, v = c2v[c];
        const uint32_t cn = k == 2 ? c - 2 : c + 1;
        pfb(fv + f); pfb(vv + v);
        fv[f] |= 1;
        proc[n] = c;
        const uint32_t gate = opp[c] != kNone ? 0x10u : 0u;
        const uint8_t vflags = vv[v];
        if (!(vflags & 1)) {
          vv[v] = vflags | 1;
          if (!(vflags & 2)) { sym[n++] = (uint8_t)(SYM_C | gate); c = opp[cn]; continue; }
        }
        const uint32_t cp = k == 0 ? c + 2 : c - 1;
        const uint32_t rc = opp[cn], lc = opp[cp];
        const uint8_t rf = rc == kNone ? 1 : fv[rc / 3], lf = lc == kNone ? 1 : fv[lc / 3];
        const bool rv = rf & 1, lv = lf & 1;
        const uint8_t nb = (uint8_t)(gate | ((rc != kNone && rv) ? 0x20u : 0u) | ((lc != kNone && lv) ? 0x40u : 0u));
        const uint64_t symbol_idx = n;
        if (rv) {
#ifndef LIB_NO_SPLITS
          if (rc != kNone && (rf & 2)) note_split(symbol_idx, 1, rc / 3);
#endif
          if (lv) {
#ifndef LIB_NO_SPLITS
            if (lc != kNone && (lf & 2)) note_split(symbol_idx, 0, lc / 3);
#endif
            sym[n++] = (uint8_t)(SYM_E | nb);
            stack.pop_back();
            break;
          }
          sym[n++] = (uint8_t)(SYM_R | nb);
          c = lc;
        } else if (lv) {
#ifndef LIB_NO_SPLITS
          if (lc != kNone && (lf & 2)) note_split(symbol_idx, 0, lc / 3);
#endif
          sym[n++] = (uint8_t)(SYM_L | nb);
          c = rc;
        } else {
          sym[n++] = (uint8_t)(SYM_S | nb);
          split_here(c, f, v, vflags, rc, lc, symbol_idx);
          break;
        }
      }
    }
    n_out = n;
  }
  double edgebreaker() {
    vvis.assign(V, 0); fvis.assign(F, 0);
    processed.reserve(F + 1); symbols.reserve(F + 1);
    const double t0 = now_ms();
    for (uint32_t f = 0; f < F && !bad; ++f) {
      if (fvis[f] & 1) continue;
      const uint32_t start = 3 * f;
      vvis[t_c2v[start]] |= 1; vvis[t_c2v[start + 1]] |= 1; vvis[t_c2v[start + 2]] |= 1;
      fvis[f] |= 1;
      run_from(t_opp[cnext(start)]);
    }
    return now_ms() - t0;
  }
};


// ---------------- shadow prefetch: the spiral's next loop runs beside its last one.  Face flags are 32-bit stamps (position in `processed` + 1), so a
// step that sees a visited neighbour knows WHERE in `processed` the previous loop passed this spot; a shadow index follows the walk one loop
// behind (re-synchronised at every such step, advanced by one otherwise), and the table lines of the face the previous loop processed D steps
// after the shadow — the neighbours of what this walk will process D steps from now — are requested into L1. ----------------
static int kShadowD = 12, kShadowMode = 0;
struct Shadow {
  uint32_t F, V;
  const uint32_t *opp, *c2v;
  uint32_t* stamp;   // per face: 0 = unvisited, else position in processed + 1 (bit 31: S face)
  uint8_t* vvis;
  uint32_t* processed; uint8_t* symbols; size_t n_processed = 0;
  std::vector<uint32_t> stack;
  void run_from(uint32_t c) {
    stack.clear(); stack.push_back(c);
    size_t q = 0;
    const int D = kShadowD;
    while (!stack.empty()) {
      c = stack.back();
      if (stamp[c / 3]) { stack.pop_back(); continue; }
      for (;;) {
        pf(opp + c); pf(c2v + c);
        const uint32_t f = c / 3, v = c2v[c];
        const size_t n = n_processed;
        {   // the previous loop, D steps ahead of where it passed this spot
          const size_t qa = q + (size_t)D;
          if (qa < n) {
            const uint32_t g = processed[qa]; __builtin_prefetch(opp + g, 0, 3); __builtin_prefetch(opp + g + 16, 0, 3); __builtin_prefetch(opp + g - 16, 0, 3); __builtin_prefetch(c2v + g, 0, 3);
            if (kShadowMode >= 1) __builtin_prefetch(stamp + g / 3, 1, 3);
            if (kShadowMode >= 2) { const uint32_t g2 = processed[q + 4]; __builtin_prefetch(vvis + c2v[g2], 1, 3); }
            if (kShadowMode >= 3) { __builtin_prefetch(c2v + g + 16, 0, 3); __builtin_prefetch(c2v + g - 16, 0, 3); }
          }
          ++q;
        }
        stamp[f] = (uint32_t)n + 1;
        processed[n] = c;
        const uint32_t gate = opp[c] != kNone ? 0x10u : 0u;
        const uint8_t vflags = vvis[v];
        if (!(vflags & 1)) {
          vvis[v] = vflags | 1;
          if (!(vflags & 2)) { symbols[n_processed++] = (uint8_t)(SYM_C | gate); c = opp[cnext(c)]; continue; }
        }
        const uint32_t rc = opp[cnext(c)], lc = opp[cprev(c)];
        const uint32_t rs = rc == kNone ? 0u : stamp[rc / 3], ls = lc == kNone ? 0u : stamp[lc / 3];
        const bool rv = rc == kNone || rs, lv = lc == kNone || ls;
        const uint8_t nb = (uint8_t)(gate | ((rc != kNone && rv) ? 0x20u : 0u) | ((lc != kNone && lv) ? 0x40u : 0u));
        if (rv) {
          if (rs) q = (rs & 0x3FFFFFFFu);          // (the right face: processed at position rs - 1; the shadow moves on from the one after it)
          if (lv) { symbols[n_processed++] = (uint8_t)(SYM_E | nb); stack.pop_back(); break; }
          symbols[n_processed++] = (uint8_t)(SYM_R | nb); c = lc;
        } else if (lv) { if (ls) q = (ls & 0x3FFFFFFFu); symbols[n_processed++] = (uint8_t)(SYM_L | nb); c = rc; }
        else { symbols[n_processed++] = (uint8_t)(SYM_S | nb); stamp[f] |= 0x80000000u; stack.back() = lc; stack.push_back(rc); break; }
      }
    }
  }

  // the sequencer with the same shadow: `order` records the corners whose face it marks, stamp2 their positions
  size_t sequence(uint32_t* seq, uint32_t* stamp2, uint32_t* order) {
    size_t n_seq = 0, n_ord = 0, q = 0;
    const int D = kShadowD;
    uint64_t left = n_processed;
    stack.clear();
    auto emit = [&](uint32_t c) { const uint32_t v = c2v[c]; if (!(vvis[v] & 4)) { vvis[v] |= 4; seq[n_seq++] = c; } };
    for (;;) {
      uint32_t c;
      if (!stack.empty()) { c = stack.back(); stack.pop_back(); }
      else if (left) { --left; c = processed[left]; }
      else break;
      if (stamp2[c / 3]) continue;
      pf(opp + c); pf(c2v + c);
      const uint32_t nc = cnext(c), pc = cprev(c);
      if (!(vvis[c2v[nc]] & 4) || !(vvis[c2v[pc]] & 4)) { emit(nc); emit(pc); stack.push_back(c); continue; }
      {
        const size_t qa = q + (size_t)D;
        if (qa < n_ord) { const uint32_t g = order[qa]; __builtin_prefetch(opp + g, 0, 3); __builtin_prefetch(opp + g + 16, 0, 3); __builtin_prefetch(opp + g - 16, 0, 3); __builtin_prefetch(c2v + g, 0, 3); }
        ++q;
      }
      stamp2[c / 3] = (uint32_t)n_ord + 1;
      order[n_ord++] = c;
      const uint32_t v = c2v[c];
      pfb(vvis + v);
      const uint32_t right = opp[nc], lft = opp[pc];
      const uint8_t vflags = vvis[v];
      if (!(vflags & 4)) {
        emit(c);
        if (!(vflags & 2)) { if (right != kNone) stack.push_back(right); continue; }
      }
      const uint32_t rs = right != kNone ? stamp2[right / 3] : 0u, ls = lft != kNone ? stamp2[lft / 3] : 0u;
      const bool rdone = rs != 0, ldone = ls != 0;
      if (rdone) { q = rs; if (!ldone && lft != kNone) stack.push_back(lft); }
      else if (ldone) { q = ls; if (right != kNone) stack.push_back(right); }
      else { if (lft != kNone) stack.push_back(lft); if (right != kNone) stack.push_back(right); }
    }
    return n_seq;
  }

  // the sequencer led by the Edgebreaker's own order: its walk runs (mostly) BACKWARDS along the traversal — half of its steps go from the face
  // processed p-th to the one processed (p-1)-th, most others hop to the neighbouring loop and go on from there — so the face the traversal
  // processed D steps before the current one is (about) what this walk reaches in D steps.  Visited flags ride in bit 30 of the stamps.
  size_t sequence_oracle(uint32_t* seq) {
    size_t n_seq = 0;
    const int D = kShadowD;
    uint64_t left = n_processed;
    stack.clear();
    constexpr uint32_t kSeq = 0x40000000u, kPos = 0x3FFFFFFFu;
    auto emit = [&](uint32_t c) { const uint32_t v = c2v[c]; if (!(vvis[v] & 4)) { vvis[v] |= 4; seq[n_seq++] = c; } };
    for (;;) {
      uint32_t c;
      if (!stack.empty()) { c = stack.back(); stack.pop_back(); }
      else if (left) { --left; c = processed[left]; }
      else break;
      const uint32_t st = stamp[c / 3];
      if (st & kSeq) continue;
      pf(opp + c); pf(c2v + c);
      const uint32_t nc = cnext(c), pc = cprev(c);
      if (!(vvis[c2v[nc]] & 4) || !(vvis[c2v[pc]] & 4)) { emit(nc); emit(pc); stack.push_back(c); continue; }
      {
        const uint32_t p = (st & kPos);          // position in processed + 1 (interior start faces: the largest value — no oracle)
        if (p > (uint32_t)D && p <= n_processed) { const uint32_t g = processed[p - 1 - D]; __builtin_prefetch(opp + g, 0, 3); __builtin_prefetch(c2v + g, 0, 3); __builtin_prefetch(stamp + g / 3, 1, 3); }
      }
      stamp[c / 3] = st | kSeq;
      const uint32_t v = c2v[c];
      pfb(vvis + v);
      const uint32_t right = opp[nc], lft = opp[pc];
      const uint8_t vflags = vvis[v];
      if (!(vflags & 4)) {
        emit(c);
        if (!(vflags & 2)) { if (right != kNone) stack.push_back(right); continue; }
      }
      const bool rdone = right != kNone && (stamp[right / 3] & kSeq), ldone = lft != kNone && (stamp[lft / 3] & kSeq);
      if (rdone) { if (!ldone && lft != kNone) stack.push_back(lft); }
      else if (ldone) { if (right != kNone) stack.push_back(right); }
      else { if (lft != kNone) stack.push_back(lft); if (right != kNone) stack.push_back(right); }
    }
    return n_seq;
  }

  // the sequencer with byte flags of its own, led by the traversal's stamps: the face the traversal processed D steps BEFORE the current one is (about)
  // what this walk reaches in D steps (it runs backwards along the traversal, or hops to the neighbouring loop and goes on from there)
  size_t sequence_oracle2(uint32_t* seq, uint8_t* fv2) {
    size_t n_seq = 0;
    const int D = kShadowD;
    uint64_t left = n_processed;
    uint32_t faces_left = F;
    stack.clear();
    auto emit = [&](uint32_t c) { const uint32_t v = c2v[c]; if (!(vvis[v] & 4)) { vvis[v] |= 4; seq[n_seq++] = c; } };
    for (;;) {
      uint32_t c;
      if (!faces_left) break;
      if (!stack.empty()) { c = stack.back(); stack.pop_back(); }
      else if (left) { --left; c = processed[left]; }
      else break;
      if (fv2[c / 3]) continue;
      pf(opp + c); pf(c2v + c);
      const uint32_t nc = cnext(c), pc = cprev(c);
      if (!(vvis[c2v[nc]] & 4) || !(vvis[c2v[pc]] & 4)) { emit(nc); emit(pc); stack.push_back(c); continue; }
      {
        const uint32_t p = stamp[c / 3] & 0x3FFFFFFFu;
        if (p > (uint32_t)D + 1 && p <= n_processed) {
          const uint32_t g = processed[p - 1 - D];
          __builtin_prefetch(opp + g, 0, 3); __builtin_prefetch(opp + g + 16, 0, 3); __builtin_prefetch(opp + g - 16, 0, 3); __builtin_prefetch(c2v + g, 0, 3);
          if (kShadowMode >= 1) __builtin_prefetch(stamp + g / 3, 0, 3);
          if (kShadowMode >= 2) __builtin_prefetch(fv2 + g / 3, 1, 3);
        }
      }
      fv2[c / 3] = 1; --faces_left;
      const uint32_t v = c2v[c];
      const uint32_t right = opp[nc], lft = opp[pc];
      const uint8_t vflags = vvis[v];
      if (!(vflags & 4)) {
        emit(c);
        if (!(vflags & 2)) { if (right != kNone) stack.push_back(right); continue; }
      }
      const bool rdone = right != kNone && fv2[right / 3], ldone = lft != kNone && fv2[lft / 3];
      if (rdone) { if (!ldone && lft != kNone) stack.push_back(lft); }
      else if (ldone) { if (right != kNone) stack.push_back(right); }
      else { if (lft != kNone) stack.push_back(lft); if (right != kNone) stack.push_back(right); }
    }
    return n_seq;
  }

  // the sequencer with 32-bit VERTEX stamps (0 = not emitted, else position in `seq` + 1): a step whose tip was emitted by the previous loop knows where in
  // `seq` that loop passed, and the table lines of the corner it emitted D entries later are requested into L1
  size_t sequence_vstamp(uint32_t* seq, uint8_t* fv2, uint32_t* vst) {
    size_t n_seq = 0;
    const int D = kShadowD;
    uint64_t left = n_processed;
    uint32_t faces_left = F;
    size_t q = ~(size_t)0 >> 1;
    stack.clear();
    auto emit = [&](uint32_t c) { const uint32_t v = c2v[c]; if (!vst[v]) { seq[n_seq] = c; vst[v] = (uint32_t)++n_seq; } };
    for (;;) {
      uint32_t c;
      if (!faces_left) break;
      if (!stack.empty()) { c = stack.back(); stack.pop_back(); }
      else if (left) { --left; c = processed[left]; }
      else break;
      if (fv2[c / 3]) continue;
      pf(opp + c); pf(c2v + c);
      const uint32_t nc = cnext(c), pc = cprev(c);
      if (!vst[c2v[nc]] || !vst[c2v[pc]]) { emit(nc); emit(pc); stack.push_back(c); continue; }
      {
        const size_t qa = q + (size_t)D;
        if (qa < n_seq) { const uint32_t g = seq[qa]; __builtin_prefetch(opp + g, 0, 3); __builtin_prefetch(opp + g + 16, 0, 3); __builtin_prefetch(opp + g - 16, 0, 3); __builtin_prefetch(c2v + g, 0, 3); }
      }
      fv2[c / 3] = 1; --faces_left;
      const uint32_t v = c2v[c];
      const uint32_t right = opp[nc], lft = opp[pc];
      const uint32_t vs = vst[v];
      if (!vs) {
        emit(c);
        ++q;
        if (right != kNone) stack.push_back(right);
        continue;
      }
      q = vs;   // (the tip was emitted at position vs - 1: the shadow moves on from the entry after it)
      const bool rdone = right != kNone && fv2[right / 3], ldone = lft != kNone && fv2[lft / 3];
      if (rdone) { if (!ldone && lft != kNone) stack.push_back(lft); }
      else if (ldone) { if (right != kNone) stack.push_back(right); }
      else { if (lft != kNone) stack.push_back(lft); if (right != kNone) stack.push_back(right); }
    }
    return n_seq;
  }
  void edgebreaker() {
    n_processed = 0;
    for (uint32_t f = 0; f < F; ++f) {
      if (stamp[f]) continue;
      const uint32_t start = 3 * f;
      vvis[c2v[start]] |= 1; vvis[c2v[start + 1]] |= 1; vvis[c2v[start + 2]] |= 1;
      stamp[f] = 0x3FFFFFFFu;
      run_from(opp[cnext(start)]);
    }
  }
};


// ---------------- per-CORNER visited flags: a neighbour's state is cvis[its corner] — no division by three for the two faces across the edges ----------------
struct CornerFlags {
  uint32_t F, V;
  const uint32_t *opp, *c2v;
  uint8_t *cvis, *vvis;      // cvis: 3F bytes
  uint32_t* processed; uint8_t* symbols; size_t n_processed = 0;
  std::vector<uint32_t> stack;
  void run_from(uint32_t c) {
    stack.clear(); stack.push_back(c);
    while (!stack.empty()) {
      c = stack.back();
      if (cvis[c] & 1) { stack.pop_back(); continue; }
      for (;;) {
        pf(opp + c); pf(c2v + c);
        const uint32_t f = c / 3, k = c - 3 * f, v = c2v[c], b = c - k;
        const uint32_t cn = k == 2 ? c - 2 : c + 1;
        cvis[b] |= 1; cvis[b + 1] |= 1; cvis[b + 2] |= 1;
        processed[n_processed] = c;
        const uint32_t gate = opp[c] != kNone ? 0x10u : 0u;
        const uint8_t vflags = vvis[v];
        if (!(vflags & 1)) {
          vvis[v] = vflags | 1;
          if (!(vflags & 2)) { symbols[n_processed++] = (uint8_t)(SYM_C | gate); c = opp[cn]; continue; }
        }
        const uint32_t cp = k == 0 ? c + 2 : c - 1;
        const uint32_t rc = opp[cn], lc = opp[cp];
        const bool rv = rc == kNone || (cvis[rc] & 1), lv = lc == kNone || (cvis[lc] & 1);
        const uint8_t nb = (uint8_t)(gate | ((rc != kNone && rv) ? 0x20u : 0u) | ((lc != kNone && lv) ? 0x40u : 0u));
        if (rv) {
          if (lv) { symbols[n_processed++] = (uint8_t)(SYM_E | nb); stack.pop_back(); break; }
          symbols[n_processed++] = (uint8_t)(SYM_R | nb); c = lc;
        } else if (lv) { symbols[n_processed++] = (uint8_t)(SYM_L | nb); c = rc; }
        else { symbols[n_processed++] = (uint8_t)(SYM_S | nb); cvis[b] |= 2; cvis[b + 1] |= 2; cvis[b + 2] |= 2; stack.back() = lc; stack.push_back(rc); break; }
      }
    }
  }
  void edgebreaker() {
    n_processed = 0;
    for (uint32_t f = 0; f < F; ++f) {
      if (cvis[3 * (size_t)f] & 1) continue;
      const uint32_t start = 3 * f;
      vvis[c2v[start]] |= 1; vvis[c2v[start + 1]] |= 1; vvis[c2v[start + 2]] |= 1;
      cvis[start] |= 1; cvis[start + 1] |= 1; cvis[start + 2] |= 1;
      run_from(opp[cnext(start)]);
    }
  }
};

// ---------------- record layout: R[8f + k] = vertex of corner k (k = 0..2), R[8f + 3] = face flags, R[8f + 4 + k] = opposite corner (as 8f' + k') ----------------
static inline uint32_t rnext(uint32_t c) { return c + 1u - 3u * ((c >> 1) & 1u); }
static inline uint32_t rprev(uint32_t c) { return c - 1u + 3u * (uint32_t)((c & 3u) == 0u); }
static inline uint32_t rflag(uint32_t c) { return (c & ~7u) | 3u; }
struct Rec {
  uint32_t F, V;
  uint32_t* R;
  uint8_t* vvis;
  uint32_t* processed; uint8_t* symbols; size_t n_processed = 0;
  std::vector<uint32_t> stack;
  static inline void pfr(const uint32_t* p) { if (kPf) { __builtin_prefetch(p + 16, 1, 2); __builtin_prefetch(p - 16, 1, 2); } }
  void run_from(uint32_t c) {
    stack.clear(); stack.push_back(c);
    while (!stack.empty()) {
      c = stack.back();
      if (R[rflag(c)] & 1) { stack.pop_back(); continue; }
      for (;;) {
        pfr(R + c);
        const uint32_t v = R[c];
        pfb(vvis + v);
        R[rflag(c)] |= 1;
        processed[n_processed] = c;
        const uint32_t gate = R[c + 4] != kNone ? 0x10u : 0u;
        const uint8_t vflags = vvis[v];
        if (!(vflags & 1)) {
          vvis[v] = vflags | 1;
          if (!(vflags & 2)) { symbols[n_processed++] = (uint8_t)(SYM_C | gate); c = R[rnext(c) + 4]; continue; }
        }
        const uint32_t rc = R[rnext(c) + 4], lc = R[rprev(c) + 4];
        const bool rv = rc == kNone || (R[rflag(rc)] & 1), lv = lc == kNone || (R[rflag(lc)] & 1);
        const uint8_t nb = (uint8_t)(gate | ((rc != kNone && rv) ? 0x20u : 0u) | ((lc != kNone && lv) ? 0x40u : 0u));
        if (rv) {
          if (lv) { symbols[n_processed++] = (uint8_t)(SYM_E | nb); stack.pop_back(); break; }
          symbols[n_processed++] = (uint8_t)(SYM_R | nb); c = lc;
        } else if (lv) { symbols[n_processed++] = (uint8_t)(SYM_L | nb); c = rc; }
        else { symbols[n_processed++] = (uint8_t)(SYM_S | nb); R[rflag(c)] |= 2; stack.back() = lc; stack.push_back(rc); break; }
      }
    }
  }
  void edgebreaker() {
    n_processed = 0;
    for (uint32_t f = 0; f < F; ++f) {
      if (R[8 * (size_t)f + 3] & 1) continue;
      const uint32_t start = 8 * f;
      vvis[R[start]] |= 1; vvis[R[start + 1]] |= 1; vvis[R[start + 2]] |= 1;
      R[start + 3] |= 1;
      run_from(R[rnext(start) + 4]);
    }
  }
  size_t sequence(uint32_t* seq) {
    size_t n_seq = 0;
    uint64_t left = n_processed;
    stack.clear();
    auto emit = [&](uint32_t c) { const uint32_t v = R[c]; if (!(vvis[v] & 4)) { vvis[v] |= 4; seq[n_seq++] = c; } };
    for (;;) {
      uint32_t c;
      if (!stack.empty()) { c = stack.back(); stack.pop_back(); }
      else if (left) { --left; c = processed[left]; }
      else break;
      if (R[rflag(c)] & 4) continue;
      pfr(R + c);
      const uint32_t nc = rnext(c), pc = rprev(c);
      if (!(vvis[R[nc]] & 4) || !(vvis[R[pc]] & 4)) { emit(nc); emit(pc); stack.push_back(c); continue; }
      R[rflag(c)] |= 4;
      const uint32_t v = R[c];
      pfb(vvis + v);
      const uint32_t right = R[nc + 4], lft = R[pc + 4];
      const uint8_t vflags = vvis[v];
      if (!(vflags & 4)) {
        emit(c);
        if (!(vflags & 2)) { if (right != kNone) stack.push_back(right); continue; }
      }
      const bool rdone = right != kNone && (R[rflag(right)] & 4), ldone = lft != kNone && (R[rflag(lft)] & 4);
      if (rdone) { if (!ldone && lft != kNone) stack.push_back(lft); }
      else if (ldone) { if (right != kNone) stack.push_back(right); }
      else { if (lft != kNone) stack.push_back(lft); if (right != kNone) stack.push_back(right); }
    }
    return n_seq;
  }
};

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 2236u;
  const int repeats = argc > 2 ? std::atoi(argv[2]) : 5;
  if (const char* e = std::getenv("DMI_PF")) kPf = std::atoi(e);
  const uint32_t F = 2 * n * n, V = n * n;
  const size_t C = 3 * (size_t)F;
  uint32_t* c2v = huge_alloc<uint32_t>(C + 64) + 32;
  uint32_t* opp = huge_alloc<uint32_t>(C + 64) + 32;
  // the torus grid of draco-oxide_amd/synth.py: quad (a, b): faces 2q = (i00, i10, i11), 2q + 1 = (i00, i11, i01)
  for (uint32_t a = 0; a < n; ++a)
    for (uint32_t b = 0; b < n; ++b) {
      const uint32_t a1 = (a + 1) % n, b1 = (b + 1) % n, q = a * n + b;
      const uint32_t i00 = a * n + b, i10 = a1 * n + b, i01 = a * n + b1, i11 = a1 * n + b1;
      uint32_t* f0 = c2v + 6 * (size_t)q;
      f0[0] = i00; f0[1] = i10; f0[2] = i11; f0[3] = i00; f0[4] = i11; f0[5] = i01;
    }
  {   // half-edge matching through a sort of (min, max, corner)
    struct E { uint64_t key; uint32_t c; };
    std::vector<E> es(C);
    for (size_t c = 0; c < C; ++c) {
      const uint32_t s = c2v[cnext((uint32_t)c)], t = c2v[cprev((uint32_t)c)];
      es[c] = {((uint64_t)std::min(s, t) << 32) | std::max(s, t), (uint32_t)c};
    }
    std::sort(es.begin(), es.end(), [](const E& x, const E& y) { return x.key < y.key; });
    for (size_t c = 0; c < C; ++c) opp[c] = kNone;
    for (size_t i = 0; i + 1 < C; ++i) if (es[i].key == es[i + 1].key) { opp[es[i].c] = es[i + 1].c; opp[es[i + 1].c] = es[i].c; ++i; }
  }
  uint32_t* processedA = huge_alloc<uint32_t>(F + 16);
  uint32_t* processedB = huge_alloc<uint32_t>(F + 16);
  uint8_t* symA = huge_alloc<uint8_t>(F + 16);
  uint8_t* symB = huge_alloc<uint8_t>(F + 16);
  uint32_t* seqA = huge_alloc<uint32_t>(V + 16);
  uint32_t* seqB = huge_alloc<uint32_t>(V + 16);
  uint8_t* fvis = huge_alloc<uint8_t>(F + 1024) + 512;
  uint8_t* vvis = huge_alloc<uint8_t>(V + 1024) + 512;
  uint32_t* R = huge_alloc<uint32_t>(8 * (size_t)F + 64) + 32;   // (64-byte aligned: mmap + 128 bytes)
  auto to_rec = [](uint32_t c) { return c == kNone ? kNone : 8u * (c / 3u) + c % 3u; };
  auto from_rec = [](uint32_t c) { return 3u * (c >> 3) + (c & 3u); };
  double best[4] = {1e30, 1e30, 1e30, 1e30};
  size_t nA = 0, nB = 0, sA = 0, sB = 0;
  for (int r = 0; r < repeats; ++r) {
    std::memset(fvis, 0, F); std::memset(vvis, 0, V);
    Flat a{F, V, opp, c2v, fvis, vvis, processedA, symA};
    double t0 = now_ms(); a.edgebreaker(); double t1 = now_ms(); sA = a.sequence(seqA); double t2 = now_ms();
    nA = a.n_processed;
    best[0] = std::min(best[0], t1 - t0); best[1] = std::min(best[1], t2 - t1);
    for (uint32_t f = 0; f < F; ++f) {
      uint32_t* rec = R + 8 * (size_t)f;
      for (int k = 0; k < 3; ++k) { rec[k] = c2v[3 * (size_t)f + k]; rec[4 + k] = to_rec(opp[3 * (size_t)f + k]); }
      rec[3] = 0; rec[7] = 0;
    }
    std::memset(vvis, 0, V);
    Rec b{F, V, R, vvis, processedB, symB};
    t0 = now_ms(); b.edgebreaker(); t1 = now_ms(); sB = b.sequence(seqB); t2 = now_ms();
    nB = b.n_processed;
    best[2] = std::min(best[2], t1 - t0); best[3] = std::min(best[3], t2 - t1);
  }
  bool same = nA == nB && sA == sB;
  for (size_t i = 0; same && i < nA; ++i) same = processedA[i] == from_rec(processedB[i]) && symA[i] == symB[i];
  for (size_t i = 0; same && i < sA; ++i) same = seqA[i] == from_rec(seqB[i]);
  size_t nsym[5] = {0, 0, 0, 0, 0};
  for (size_t i = 0; i < nA; ++i) ++nsym[symA[i] & 7];
  std::printf("torus n=%u: F=%u processed=%zu seq=%zu  C/S/L/R/E = %zu/%zu/%zu/%zu/%zu  same order: %s  (DMI_PF=%d)\n", n, F, nA, sA, nsym[0], nsym[1], nsym[2], nsym[3], nsym[4], same ? "yes" : "NO", kPf);
  std::printf("  separate arrays : Edgebreaker %.1f ms (%.2f ns/face), sequencer %.1f ms (%.2f ns/face)\n", best[0], best[0] * 1e6 / F, best[1], best[1] * 1e6 / F);
  std::printf("  32-byte records : Edgebreaker %.1f ms (%.2f ns/face), sequencer %.1f ms (%.2f ns/face)\n", best[2], best[2] * 1e6 / F, best[3], best[3] * 1e6 / F);
  {
    uint32_t* RL = huge_alloc<uint32_t>(2 * C + 64) + 32;
    uint32_t* cv = huge_alloc<uint32_t>(C + 64) + 32;
    for (size_t c = 0; c < C; ++c) { RL[2 * c] = opp[cnext((uint32_t)c)]; RL[2 * c + 1] = opp[cprev((uint32_t)c)]; cv[c] = c2v[c] | (opp[c] != kNone ? 0x80000000u : 0u); }
    double b0 = 1e30, b1 = 1e30; size_t nH = 0, sH = 0;
    for (int r = 0; r < repeats; ++r) {
      std::memset(fvis, 0, F); std::memset(vvis, 0, V);
      HopTables h{F, V, RL, cv, fvis, vvis, processedB, symB};
      double t0 = now_ms(); h.edgebreaker(); double t1 = now_ms(); sH = h.sequence(seqB); double t2 = now_ms();
      nH = h.n_processed; b0 = std::min(b0, t1 - t0); b1 = std::min(b1, t2 - t1);
    }
    bool ok = nH == nA && sH == sA;
    for (size_t i = 0; ok && i < nA; ++i) ok = processedA[i] == processedB[i] && symA[i] == symB[i];
    for (size_t i = 0; ok && i < sA; ++i) ok = seqA[i] == seqB[i];
    std::printf("  right/left tables: Edgebreaker %.1f ms (%.2f ns/face), sequencer %.1f ms (%.2f ns/face)  same order: %s\n", b0, b0 * 1e6 / F, b1, b1 * 1e6 / F, ok ? "yes" : "NO");
  }
  {
    uint32_t* stamp = huge_alloc<uint32_t>(F + 64) + 32;
    uint32_t* stamp2 = huge_alloc<uint32_t>(F + 64) + 32;
    uint32_t* order = huge_alloc<uint32_t>(F + 64) + 32;
    for (int mode = 0; mode < (std::getenv("SEQ_O2") ? 3 : 1); ++mode) for (int D : {0, 8, 16, 32}) {
      kShadowD = D; kShadowMode = mode;
      double b0 = 1e30, bs = 1e30; size_t nS = 0, sS = 0;
      for (int r = 0; r < repeats; ++r) {
        std::memset(stamp, 0, 4 * (size_t)F); std::memset(vvis, 0, V);
        Shadow h{F, V, opp, c2v, stamp, vvis, processedB, symB};
        double t0 = now_ms(); h.edgebreaker(); double t1 = now_ms();
        nS = h.n_processed; b0 = std::min(b0, t1 - t0);
        if (std::getenv("SEQ_VS")) { std::memset(fvis, 0, F); std::memset(stamp2, 0, 4 * (size_t)V); t0 = now_ms(); sS = h.sequence_vstamp(seqB, fvis, stamp2); t1 = now_ms(); }
        else if (std::getenv("SEQ_O2")) { std::memset(fvis, 0, F); t0 = now_ms(); sS = h.sequence_oracle2(seqB, fvis); t1 = now_ms(); }
        else if (std::getenv("SEQ_OWN")) { std::memset(stamp2, 0, 4 * (size_t)F); t0 = now_ms(); sS = h.sequence(seqB, stamp2, order); t1 = now_ms(); }
        else { t0 = now_ms(); sS = h.sequence_oracle(seqB); t1 = now_ms(); }
        bs = std::min(bs, t1 - t0);
      }
      bool ok = nS == nA;
      for (size_t i = 0; ok && i < nA; ++i) ok = processedA[i] == processedB[i] && symA[i] == symB[i];
      ok = ok && sS == sA;
      for (size_t i = 0; ok && i < sA; ++i) ok = seqA[i] == seqB[i];
      std::printf("  shadow L1 prefetch, mode %d D = %2d: Edgebreaker %.1f ms (%.2f ns/face), sequencer %.1f ms (%.2f ns/face)  same order: %s\n", mode, D, b0, b0 * 1e6 / F, bs, bs * 1e6 / F, ok ? "yes" : "NO");
    }
  }
  {
    uint8_t* cvis = huge_alloc<uint8_t>(C + 1024) + 512;
    double b0 = 1e30; size_t nC = 0;
    for (int r = 0; r < repeats; ++r) {
      std::memset(cvis, 0, C); std::memset(vvis, 0, V);
      CornerFlags h{F, V, opp, c2v, cvis, vvis, processedB, symB};
      double t0 = now_ms(); h.edgebreaker(); double t1 = now_ms();
      nC = h.n_processed; b0 = std::min(b0, t1 - t0);
    }
    bool ok = nC == nA;
    for (size_t i = 0; ok && i < nA; ++i) ok = processedA[i] == processedB[i] && symA[i] == symB[i];
    std::printf("  per-corner visited flags: Edgebreaker %.1f ms (%.2f ns/face)  same order: %s\n", b0, b0 * 1e6 / F, ok ? "yes" : "NO");
  }
  { double best = 1e30; size_t nn = 0; for (int r = 0; r < repeats; ++r) { LibWalker w{F, V, opp, c2v}; best = std::min(best, w.edgebreaker()); nn = w.n_out; }
    std::printf("  the library's lean loop, verbatim: %.1f ms (%.2f ns/face), %zu symbols\n", best, best * 1e6 / F, nn); }
  run_prod<true, true, true>("push_back + guards + splits", F, V, opp, c2v, processedB, symB, repeats);
  run_prod<false, true, true>("raw outputs + guards + splits", F, V, opp, c2v, processedB, symB, repeats);
  run_prod<true, false, true>("push_back, no guards, splits", F, V, opp, c2v, processedB, symB, repeats);
  run_prod<true, true, false>("push_back + guards, no splits", F, V, opp, c2v, processedB, symB, repeats);
  run_prod<false, false, false>("raw, no guards, no splits", F, V, opp, c2v, processedB, symB, repeats);
  return same ? 0 : 1;
}
