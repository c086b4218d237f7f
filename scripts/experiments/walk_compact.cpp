// walk_compact.cpp — VERDICT r4 #8, the one bounded experiment on the single-mesh floor: do the two serial host walks (Edgebreaker traversal,
// attribute sequencer) step faster over COMPACT corner tables — opposite corners as 16-bit deltas (opp[c] - c, an escape value sends the rare
// far ones to the 32-bit table), vertices as 24-bit ids (3 bytes per corner) — than over the 2 × 32-bit arrays of host_conn.cpp?  5 bytes per
// corner instead of 8: 150 MB instead of 240 MB for the 10M-triangle workload.  Same loops (the plain ones of walk_layout.cpp's `Flat`), same
// visiting order (checked).  CPU only:  g++ -O2 -std=c++17 -o walk_compact.out walk_compact.cpp && ./walk_compact.out [n=2236] [repeats=5]
#include <sys/mman.h>

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
template <class T>
static T* huge_alloc(size_t n) {
  const size_t bytes = ((n * sizeof(T) + (2u << 20) - 1) >> 21) << 21;
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { std::perror("mmap"); std::exit(1); }
  madvise(p, bytes, MADV_HUGEPAGE);
  std::memset(p, 0, bytes);
  return static_cast<T*>(p);
}
static int kPf = 16;
enum : uint8_t { SYM_C, SYM_S, SYM_L, SYM_R, SYM_E };

struct Wide {   // the production layout
  const uint32_t *opp_, *c2v_;
  inline uint32_t opp(uint32_t c) const { return opp_[c]; }
  inline uint32_t c2v(uint32_t c) const { return c2v_[c]; }
  inline void pf(uint32_t c) const { if (kPf) { __builtin_prefetch(opp_ + c + kPf, 0, 2); __builtin_prefetch(opp_ + c - kPf, 0, 2); __builtin_prefetch(c2v_ + c + kPf, 0, 2); __builtin_prefetch(c2v_ + c - kPf, 0, 2); } }
};
struct Compact {
  const int16_t* d16; const uint32_t* opp32; const uint8_t* v24;
  inline uint32_t opp(uint32_t c) const { const int16_t d = d16[c]; return d == INT16_MIN ? opp32[c] : (uint32_t)((int32_t)c + d); }
  inline uint32_t c2v(uint32_t c) const { uint32_t w; std::memcpy(&w, v24 + 3 * (size_t)c, 4); return w & 0xFFFFFFu; }
  inline void pf(uint32_t c) const { if (kPf) { __builtin_prefetch(d16 + c + 2 * kPf, 0, 2); __builtin_prefetch(d16 + c - 2 * kPf, 0, 2); __builtin_prefetch(v24 + 3 * (size_t)c + 4 * kPf, 0, 2); __builtin_prefetch(v24 + 3 * (size_t)c - 4 * kPf, 0, 2); } }
};

template <class T>
struct Walk {
  uint32_t F, V;
  T t;
  uint8_t *fvis, *vvis;
  uint32_t* processed; uint8_t* symbols; size_t n_processed = 0;
  std::vector<uint32_t> stack;
  void run_from(uint32_t c) {
    stack.clear(); stack.push_back(c);
    while (!stack.empty()) {
      c = stack.back();
      if (fvis[c / 3] & 1) { stack.pop_back(); continue; }
      for (;;) {
        t.pf(c);
        const uint32_t f = c / 3, v = t.c2v(c);
        fvis[f] |= 1;
        processed[n_processed] = c;
        const uint32_t gate = t.opp(c) != kNone ? 0x10u : 0u;
        const uint8_t vflags = vvis[v];
        if (!(vflags & 1)) {
          vvis[v] = vflags | 1;
          if (!(vflags & 2)) { symbols[n_processed++] = (uint8_t)(SYM_C | gate); c = t.opp(cnext(c)); continue; }
        }
        const uint32_t rc = t.opp(cnext(c)), lc = t.opp(cprev(c));
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
      vvis[t.c2v(start)] |= 1; vvis[t.c2v(start + 1)] |= 1; vvis[t.c2v(start + 2)] |= 1;
      fvis[f] |= 1;
      run_from(t.opp(cnext(start)));
    }
  }
  size_t sequence(uint32_t* seq) {
    size_t n_seq = 0;
    uint64_t left = n_processed;
    stack.clear();
    auto emit = [&](uint32_t c) { const uint32_t v = t.c2v(c); if (!(vvis[v] & 4)) { vvis[v] |= 4; seq[n_seq++] = c; } };
    for (;;) {
      uint32_t c;
      if (!stack.empty()) { c = stack.back(); stack.pop_back(); }
      else if (left) { --left; c = processed[left]; }
      else break;
      if (fvis[c / 3] & 4) continue;
      t.pf(c);
      const uint32_t nc = cnext(c), pc = cprev(c);
      if (!(vvis[t.c2v(nc)] & 4) || !(vvis[t.c2v(pc)] & 4)) { emit(nc); emit(pc); stack.push_back(c); continue; }
      fvis[c / 3] |= 4;
      const uint32_t v = t.c2v(c);
      const uint32_t right = t.opp(nc), lft = t.opp(pc);
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

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 2236u;
  const int repeats = argc > 2 ? std::atoi(argv[2]) : 5;
  if (const char* e = std::getenv("DMI_PF")) kPf = std::atoi(e);
  const uint32_t F = 2 * n * n, V = n * n;
  const size_t C = 3 * (size_t)F;
  uint32_t* c2v = huge_alloc<uint32_t>(C + 64) + 32;
  uint32_t* opp = huge_alloc<uint32_t>(C + 64) + 32;
  for (uint32_t a = 0; a < n; ++a)
    for (uint32_t b = 0; b < n; ++b) {
      const uint32_t a1 = (a + 1) % n, b1 = (b + 1) % n, q = a * n + b;
      const uint32_t i00 = a * n + b, i10 = a1 * n + b, i01 = a * n + b1, i11 = a1 * n + b1;
      uint32_t* f0 = c2v + 6 * (size_t)q;
      f0[0] = i00; f0[1] = i10; f0[2] = i11; f0[3] = i00; f0[4] = i11; f0[5] = i01;
    }
  {
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
  int16_t* d16 = huge_alloc<int16_t>(C + 128) + 64;
  uint8_t* v24 = huge_alloc<uint8_t>(3 * C + 256) + 128;
  size_t escapes = 0;
  const double tc0 = now_ms();
  for (size_t c = 0; c < C; ++c) {
    const int64_t d = opp[c] == kNone ? (int64_t)1 << 40 : (int64_t)opp[c] - (int64_t)c;
    if (d > INT16_MIN && d <= INT16_MAX) d16[c] = (int16_t)d; else { d16[c] = INT16_MIN; ++escapes; }
    std::memcpy(v24 + 3 * c, &c2v[c], 3);
  }
  const double t_convert = now_ms() - tc0;
  uint32_t *pA = huge_alloc<uint32_t>(F + 16), *pB = huge_alloc<uint32_t>(F + 16), *sqA = huge_alloc<uint32_t>(V + 16), *sqB = huge_alloc<uint32_t>(V + 16);
  uint8_t *syA = huge_alloc<uint8_t>(F + 16), *syB = huge_alloc<uint8_t>(F + 16);
  uint8_t* fvis = huge_alloc<uint8_t>(F + 1024) + 512;
  uint8_t* vvis = huge_alloc<uint8_t>(V + 1024) + 512;
  double best[4] = {1e30, 1e30, 1e30, 1e30};
  size_t nA = 0, nB = 0, sA = 0, sB = 0;
  for (int r = 0; r < repeats; ++r) {
    std::memset(fvis, 0, F); std::memset(vvis, 0, V);
    Walk<Wide> a{F, V, Wide{opp, c2v}, fvis, vvis, pA, syA};
    double t0 = now_ms(); a.edgebreaker(); double t1 = now_ms(); sA = a.sequence(sqA); double t2 = now_ms();
    nA = a.n_processed; best[0] = std::min(best[0], t1 - t0); best[1] = std::min(best[1], t2 - t1);
    std::memset(fvis, 0, F); std::memset(vvis, 0, V);
    Walk<Compact> b{F, V, Compact{d16, opp, v24}, fvis, vvis, pB, syB};
    t0 = now_ms(); b.edgebreaker(); t1 = now_ms(); sB = b.sequence(sqB); t2 = now_ms();
    nB = b.n_processed; best[2] = std::min(best[2], t1 - t0); best[3] = std::min(best[3], t2 - t1);
  }
  bool same = nA == nB && sA == sB;
  for (size_t i = 0; same && i < nA; ++i) same = pA[i] == pB[i] && syA[i] == syB[i];
  for (size_t i = 0; same && i < sA; ++i) same = sqA[i] == sqB[i];
  std::printf("torus n=%u: F=%u  same order: %s  (DMI_PF=%d)  escapes to the 32-bit table: %zu of %zu corners; conversion of the tables %.1f ms on one core\n", n, F, same ? "yes" : "NO", kPf, escapes, C, t_convert);
  std::printf("  2 x 32-bit arrays (8 B/corner)          : Edgebreaker %.1f ms (%.2f ns/face), sequencer %.1f ms (%.2f ns/face)\n", best[0], best[0] * 1e6 / F, best[1], best[1] * 1e6 / F);
  std::printf("  16-bit deltas + 24-bit ids (5 B/corner) : Edgebreaker %.1f ms (%.2f ns/face), sequencer %.1f ms (%.2f ns/face)\n", best[2], best[2] * 1e6 / F, best[3], best[3] * 1e6 / F);
  return same ? 0 : 1;
}
