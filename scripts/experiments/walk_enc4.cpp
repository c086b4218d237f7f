// walk_enc4.cpp — corner ids as 4·face + k instead of 3·face + k in the tables the two serial walks chase (round 5).  A step of a walk goes
// corner → face (c / 3: a multiply-high and a shift on the dependency chain) → k = c − 3·face → next corner → opposite[next] → …; with
// ids 4·face + k the face is a shift and k a mask.  The arrays stay dense (three entries per face, index = c − (c >> 2)); only the VALUES stored in
// `opposite` (and what the walk emits) change their encoding.  Same visiting order in both forms (checked).  CPU only.
//   g++ -O2 -std=c++17 -o walk_enc4.out walk_enc4.cpp && ./walk_enc4.out [n=2236] [repeats=5]
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/mman.h>

static constexpr uint32_t kNone = 0xFFFFFFFFu;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class T> static T* huge_alloc(size_t n) {
  const size_t bytes = ((n * sizeof(T) + (2u << 20) - 1) >> 21) << 21;
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { std::perror("mmap"); std::exit(1); }
  madvise(p, bytes, MADV_HUGEPAGE);
  std::memset(p, 0, bytes);
  return static_cast<T*>(p);
}
static inline void pf(const uint32_t* p) { __builtin_prefetch(p + 16, 0, 2); __builtin_prefetch(p - 16, 0, 2); }

struct E3 {
  static uint32_t face(uint32_t c) { return c / 3; }
  static uint32_t k(uint32_t c, uint32_t f) { return c - 3 * f; }
  static uint32_t idx(uint32_t c, uint32_t) { return c; }
  static uint32_t enc(uint32_t c3) { return c3; }
  static uint32_t dec(uint32_t c) { return c; }
};
struct E4 {
  static uint32_t face(uint32_t c) { return c >> 2; }
  static uint32_t k(uint32_t c, uint32_t) { return c & 3u; }
  static uint32_t idx(uint32_t c, uint32_t f) { return c - f; }
  static uint32_t enc(uint32_t c3) { return c3 == kNone ? kNone : c3 + c3 / 3; }
  static uint32_t dec(uint32_t c) { return c - (c >> 2); }
};
enum : uint8_t { SYM_C, SYM_S, SYM_L, SYM_R, SYM_E };

// the Edgebreaker traversal of host_conn.cpp's Walker::run_from_t<true> (stamps + the shadow prefetch one loop behind), boundary-free mesh
template <class E>
struct Eb {
  uint32_t F, V;
  const uint32_t *opp, *c2v;     // opp: values in E's encoding; both indexed densely (3 per face)
  uint32_t* st; uint8_t* vv;
  uint32_t* proc; uint8_t* sym; size_t n = 0;
  std::vector<uint32_t> stack;
  void run_from(uint32_t c) {
    size_t q = ~(size_t)0 >> 1;
    constexpr size_t kAhead = 12;
    stack.clear(); stack.push_back(c);
    while (!stack.empty()) {
      c = stack.back();
      if (st[E::face(c)] != 0u) { stack.pop_back(); continue; }
      for (;;) {
        const uint32_t f = E::face(c), k = E::k(c, f), i = E::idx(c, f);
        pf(opp + i); pf(c2v + i);
        const uint32_t v = c2v[i];
        const uint32_t cn = k == 2 ? c - 2 : c + 1;
        { const size_t qa = q + kAhead; if (qa < n) { const uint32_t g = E::dec(proc[qa]); __builtin_prefetch(opp + g, 0, 3); __builtin_prefetch(opp + g + 16, 0, 3); __builtin_prefetch(opp + g - 16, 0, 3); __builtin_prefetch(c2v + g, 0, 3); } ++q; }
        st[f] = (uint32_t)n + 1u;
        proc[n] = c;
        const uint8_t vflags = vv[v];
        if (!(vflags & 1)) { vv[v] = vflags | 1; sym[n++] = SYM_C; c = opp[E::idx(cn, f)]; continue; }
        const uint32_t cp = k == 0 ? c + 2 : c - 1;
        const uint32_t rc = opp[E::idx(cn, f)], lc = opp[E::idx(cp, f)];
        const uint32_t rs = st[E::face(rc)], ls = st[E::face(lc)];
        if (rs) {
          q = (size_t)(rs & 0x7FFFFFFFu);
          if (ls) { sym[n++] = SYM_E; stack.pop_back(); break; }
          sym[n++] = SYM_R; c = lc;
        } else if (ls) {
          q = (size_t)(ls & 0x7FFFFFFFu);
          sym[n++] = SYM_L; c = rc;
        } else {
          sym[n++] = SYM_S; stack.back() = lc; stack.push_back(rc); break;
        }
      }
    }
  }
};

// the same traversal over HOP tables: R[i] = opposite[next(c)], L[i] = opposite[prev(c)] for the corner whose entries are at i (ids 4·face + k): the next
// corner of a step is ONE load away from the current one (shift, subtract, load) instead of mask, compare, select, subtract, load
struct EbHops {
  uint32_t F, V;
  const uint32_t *R, *L, *c2v;
  uint32_t* st; uint8_t* vv;
  uint32_t* proc; uint8_t* sym; size_t n = 0;
  std::vector<uint32_t> stack;
  void run_from(uint32_t c) {
    size_t q = ~(size_t)0 >> 1;
    constexpr size_t kAhead = 12;
    stack.clear(); stack.push_back(c);
    while (!stack.empty()) {
      c = stack.back();
      if (st[c >> 2] != 0u) { stack.pop_back(); continue; }
      for (;;) {
        const uint32_t f = c >> 2, i = c - f;
        pf(R + i); pf(c2v + i);
        const uint32_t v = c2v[i];
        const uint32_t rc = R[i];
        { const size_t qa = q + kAhead; if (qa < n) { const uint32_t g = proc[qa]; const uint32_t gi = g - (g >> 2); __builtin_prefetch(R + gi, 0, 3); __builtin_prefetch(R + gi + 16, 0, 3); __builtin_prefetch(R + gi - 16, 0, 3); __builtin_prefetch(c2v + gi, 0, 3); } ++q; }
        st[f] = (uint32_t)n + 1u;
        proc[n] = c;
        const uint8_t vflags = vv[v];
        if (!(vflags & 1)) { vv[v] = vflags | 1; sym[n++] = SYM_C; c = rc; continue; }
        const uint32_t lc = L[i];
        const uint32_t rs = st[rc >> 2], ls = st[lc >> 2];
        if (rs) {
          q = (size_t)(rs & 0x7FFFFFFFu);
          if (ls) { sym[n++] = SYM_E; stack.pop_back(); break; }
          sym[n++] = SYM_R; c = lc;
        } else if (ls) {
          q = (size_t)(ls & 0x7FFFFFFFu);
          sym[n++] = SYM_L; c = rc;
        } else {
          sym[n++] = SYM_S; stack.back() = lc; stack.push_back(rc); break;
        }
      }
    }
  }
};

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 2236u;
  const int repeats = argc > 2 ? std::atoi(argv[2]) : 5;
  const uint32_t F = 2 * n * n, V = n * n;
  const size_t C = 3 * (size_t)F;
  uint32_t* c2v = huge_alloc<uint32_t>(C + 64) + 32;
  uint32_t* opp3 = huge_alloc<uint32_t>(C + 64) + 32;
  uint32_t* opp4 = huge_alloc<uint32_t>(C + 64) + 32;
  for (uint32_t a = 0; a < n; ++a)
    for (uint32_t b = 0; b < n; ++b) {
      const uint32_t a1 = (a + 1) % n, b1 = (b + 1) % n, q = a * n + b;
      const uint32_t i00 = a * n + b, i10 = a1 * n + b, i01 = a * n + b1, i11 = a1 * n + b1;
      uint32_t* f0 = c2v + 6 * (size_t)q;
      f0[0] = i00; f0[1] = i10; f0[2] = i11; f0[3] = i00; f0[4] = i11; f0[5] = i01;
    }
  {
    struct Ed { uint64_t key; uint32_t c; };
    std::vector<Ed> es(C);
    auto nx = [](uint32_t c) { return c % 3 == 2 ? c - 2 : c + 1; };
    auto pv = [](uint32_t c) { return c % 3 == 0 ? c + 2 : c - 1; };
    for (size_t c = 0; c < C; ++c) { const uint32_t s = c2v[nx((uint32_t)c)], t = c2v[pv((uint32_t)c)]; es[c] = {((uint64_t)std::min(s, t) << 32) | std::max(s, t), (uint32_t)c}; }
    std::sort(es.begin(), es.end(), [](const Ed& x, const Ed& y) { return x.key < y.key; });
    for (size_t c = 0; c < C; ++c) opp3[c] = kNone;
    for (size_t i = 0; i + 1 < C; ++i) if (es[i].key == es[i + 1].key) { opp3[es[i].c] = es[i + 1].c; opp3[es[i + 1].c] = es[i].c; ++i; }
    for (size_t c = 0; c < C; ++c) opp4[c] = E4::enc(opp3[c]);
  }
  uint32_t* hopR = huge_alloc<uint32_t>(C + 64) + 32;
  uint32_t* hopL = huge_alloc<uint32_t>(C + 64) + 32;
  for (size_t c = 0; c < C; ++c) { const uint32_t k = (uint32_t)(c % 3); hopR[c] = opp4[k == 2 ? c - 2 : c + 1]; hopL[c] = opp4[k == 0 ? c + 2 : c - 1]; }
  uint32_t* st = huge_alloc<uint32_t>(F + 64) + 32;
  uint8_t* vv = huge_alloc<uint8_t>(V + 256) + 128;
  uint32_t* proc = huge_alloc<uint32_t>(F + 64);
  uint8_t* sym = huge_alloc<uint8_t>(F + 64);
  std::vector<uint32_t> ref;
  for (int r = 0; r < repeats; ++r) {
    for (int form = 0; form < 3; ++form) {
      std::memset(st, 0, (size_t)F * 4); std::memset(vv, 0, V);
      const double t0 = now_ms();
      size_t done = 0;
      if (form == 0) {
        Eb<E3> w{F, V, opp3, c2v, st, vv, proc, sym};
        // start like run_edgebreaker on a closed mesh: face 0 is the start face, the walk begins across its first edge
        st[0] = 0x7FFFFFFFu; vv[c2v[0]] |= 1; vv[c2v[1]] |= 1; vv[c2v[2]] |= 1;
        w.run_from(opp3[1]);
        done = w.n;
      } else if (form == 1) {
        Eb<E4> w{F, V, opp4, c2v, st, vv, proc, sym};
        st[0] = 0x7FFFFFFFu; vv[c2v[0]] |= 1; vv[c2v[1]] |= 1; vv[c2v[2]] |= 1;
        w.run_from(opp4[1]);
        done = w.n;
      } else {
        EbHops w{F, V, hopR, hopL, c2v, st, vv, proc, sym};
        st[0] = 0x7FFFFFFFu; vv[c2v[0]] |= 1; vv[c2v[1]] |= 1; vv[c2v[2]] |= 1;
        w.run_from(opp4[1]);
        done = w.n;
      }
      const double t1 = now_ms();
      bool same = true;
      if (form == 0) ref.assign(proc, proc + done);
      else { same = done == ref.size(); for (size_t i = 0; i < done && same; ++i) same = E4::dec(proc[i]) == ref[i]; }
      std::printf("%s: %zu faces in %.1f ms (%.2f ns per face)%s\n", form == 0 ? "3f+k" : (form == 1 ? "4f+k" : "hops"), done, t1 - t0, (t1 - t0) * 1e6 / (double)done, form >= 1 ? (same ? "  [same order]" : "  [ORDER DIFFERS]") : "");
    }
  }
  return 0;
}
