// walk_interleave.cpp — two INDEPENDENT Edgebreaker traversals (two meshes of a batch) stepped alternately by one thread against one after the other:
// each walk is a dependency chain of one cache hit per step, so a core has issue slots left for a second chain.  CPU only.
//   g++ -O2 -std=c++17 -o walk_interleave.out walk_interleave.cpp && ./walk_interleave.out [n=300] [repeats=20]
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static constexpr uint32_t kNone = 0xFFFFFFFFu;
static inline uint32_t cnext(uint32_t c) { return (c % 3 == 2) ? c - 2 : c + 1; }
static inline uint32_t cprev(uint32_t c) { return (c % 3 == 0) ? c + 2 : c - 1; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
enum : uint8_t { SYM_C, SYM_S, SYM_L, SYM_R, SYM_E };

struct Mesh {
  uint32_t F, V;
  std::vector<uint32_t> opp, c2v;
  void torus(uint32_t n) {
    F = 2 * n * n; V = n * n;
    const size_t C = 3 * (size_t)F;
    c2v.resize(C); opp.assign(C, kNone);
    for (uint32_t a = 0; a < n; ++a)
      for (uint32_t b = 0; b < n; ++b) {
        const uint32_t a1 = (a + 1) % n, b1 = (b + 1) % n, q = a * n + b;
        const uint32_t i00 = a * n + b, i10 = a1 * n + b, i01 = a * n + b1, i11 = a1 * n + b1;
        uint32_t* f0 = c2v.data() + 6 * (size_t)q;
        f0[0] = i00; f0[1] = i10; f0[2] = i11; f0[3] = i00; f0[4] = i11; f0[5] = i01;
      }
    struct E { uint64_t key; uint32_t c; };
    std::vector<E> es(C);
    for (size_t c = 0; c < C; ++c) { const uint32_t s = c2v[cnext((uint32_t)c)], t = c2v[cprev((uint32_t)c)]; es[c] = {((uint64_t)std::min(s, t) << 32) | std::max(s, t), (uint32_t)c}; }
    std::sort(es.begin(), es.end(), [](const E& x, const E& y) { return x.key < y.key; });
    for (size_t i = 0; i + 1 < C; ++i) if (es[i].key == es[i + 1].key) { opp[es[i].c] = es[i + 1].c; opp[es[i + 1].c] = es[i].c; ++i; }
  }
};

// a resumable traversal: step() processes one face, false when the mesh is done
struct Walk {
  const uint32_t *opp, *c2v;
  uint32_t F;
  std::vector<uint8_t> fvis, vvis, symbols;
  std::vector<uint32_t> processed, stack;
  size_t n = 0;
  uint32_t c = kNone, scan = 0;
  bool in_run = false;
  void init(const Mesh& m) {
    opp = m.opp.data(); c2v = m.c2v.data(); F = m.F;
    fvis.assign(m.F, 0); vvis.assign(m.V, 0); symbols.resize(m.F + 1); processed.resize(m.F + 1); stack.clear(); n = 0; scan = 0; in_run = false; c = kNone;
  }
  inline bool next_start() {   // the next unvisited face becomes an interior start face
    while (scan < F && (fvis[scan] & 1)) ++scan;
    if (scan >= F) return false;
    const uint32_t start = 3 * scan;
    vvis[c2v[start]] |= 1; vvis[c2v[start + 1]] |= 1; vvis[c2v[start + 2]] |= 1;
    fvis[scan] |= 1;
    stack.clear(); stack.push_back(opp[cnext(start)]);
    c = kNone; in_run = true;
    return true;
  }
  inline bool step() {
    if (c == kNone) {                       // pick up the stack (or a new start face)
      for (;;) {
        if (stack.empty()) { if (!next_start()) return false; }
        c = stack.back();
        if (fvis[c / 3] & 1) { stack.pop_back(); c = kNone; continue; }
        break;
      }
    }
    const uint32_t f = c / 3, v = c2v[c];
    fvis[f] |= 1;
    processed[n] = c;
    const uint32_t gate = opp[c] != kNone ? 0x10u : 0u;
    const uint8_t vflags = vvis[v];
    if (!(vflags & 1)) { vvis[v] = vflags | 1; symbols[n++] = (uint8_t)(SYM_C | gate); c = opp[cnext(c)]; return true; }
    const uint32_t rc = opp[cnext(c)], lc = opp[cprev(c)];
    const bool rv = rc == kNone || (fvis[rc / 3] & 1), lv = lc == kNone || (fvis[lc / 3] & 1);
    const uint8_t nb = (uint8_t)(gate | ((rc != kNone && rv) ? 0x20u : 0u) | ((lc != kNone && lv) ? 0x40u : 0u));
    if (rv) {
      if (lv) { symbols[n++] = (uint8_t)(SYM_E | nb); stack.pop_back(); c = kNone; return true; }
      symbols[n++] = (uint8_t)(SYM_R | nb); c = lc;
    } else if (lv) { symbols[n++] = (uint8_t)(SYM_L | nb); c = rc; }
    else { symbols[n++] = (uint8_t)(SYM_S | nb); fvis[f] |= 2; stack.back() = lc; stack.push_back(rc); c = kNone; }
    return true;
  }
};

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 300u;
  const int repeats = argc > 2 ? std::atoi(argv[2]) : 20;
  Mesh m[4];
  for (int k = 0; k < 4; ++k) m[k].torus(n + 7 * k);
  Walk w[4];
  double seq = 1e30, two = 1e30, four = 1e30;
  size_t total = 0;
  std::vector<uint32_t> ref[4];
  for (int r = 0; r < repeats; ++r) {
    for (int k = 0; k < 4; ++k) w[k].init(m[k]);
    double t0 = now_ms();
    for (int k = 0; k < 4; ++k) while (w[k].step()) {}
    seq = std::min(seq, now_ms() - t0);
    total = 0; for (int k = 0; k < 4; ++k) { total += w[k].n; ref[k].assign(w[k].processed.begin(), w[k].processed.begin() + w[k].n); }
    for (int k = 0; k < 4; ++k) w[k].init(m[k]);
    t0 = now_ms();
    for (int p = 0; p < 4; p += 2) {
      bool a = true, b = true;
      while (a && b) { a = w[p].step(); b = w[p + 1].step(); }
      while (a) a = w[p].step();
      while (b) b = w[p + 1].step();
    }
    two = std::min(two, now_ms() - t0);
    for (int k = 0; k < 4; ++k) if (!std::equal(ref[k].begin(), ref[k].end(), w[k].processed.begin())) { std::printf("MISMATCH (two)\n"); return 1; }
    for (int k = 0; k < 4; ++k) w[k].init(m[k]);
    t0 = now_ms();
    {
      bool al[4] = {true, true, true, true};
      while (al[0] || al[1] || al[2] || al[3]) { for (int k = 0; k < 4; ++k) if (al[k]) al[k] = w[k].step(); }
    }
    four = std::min(four, now_ms() - t0);
    for (int k = 0; k < 4; ++k) if (!std::equal(ref[k].begin(), ref[k].end(), w[k].processed.begin())) { std::printf("MISMATCH (four)\n"); return 1; }
  }
  std::printf("4 torus meshes n=%u.. (%zu faces): one after the other %.2f ms (%.2f ns/face), two at a time %.2f ms (%.2f), four at a time %.2f ms (%.2f)\n", n, total, seq, seq * 1e6 / total, two, two * 1e6 / total, four, four * 1e6 / total);
  return 0;
}
