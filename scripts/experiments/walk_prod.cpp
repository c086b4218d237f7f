// walk_prod.cpp — the library's own run_edgebreaker / attribute_sequence (host_conn.cpp) on tables this program allocates itself (one thread,
// mmap + MADV_HUGEPAGE): separates the walks' code from where a call's tables live.  Links libdraco_mi.so.
//   g++ -O2 -std=c++17 -I../../draco-oxide_amd/csrc walk_prod.cpp -L../../draco-oxide_amd -ldraco_mi -Wl,-rpath,'$ORIGIN/../../draco-oxide_amd' -o walk_prod.out
#include <chrono>
#include <sys/mman.h>
#include "dmi_host.hpp"
using namespace dmi;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class T> static T* huge_alloc(size_t n) {
  const size_t bytes = ((n * sizeof(T) + (2u << 20) - 1) >> 21) << 21;
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  madvise(p, bytes, MADV_HUGEPAGE);
  std::memset(p, 0, bytes);
  return static_cast<T*>(p);
}
int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 2236u;
  const int repeats = argc > 2 ? std::atoi(argv[2]) : 4;
  const uint32_t F = 2 * n * n, V = n * n;
  const size_t C = 3 * (size_t)F;
  uint32_t* c2v = huge_alloc<uint32_t>(C + 64) + 32;
  uint32_t* opp = huge_alloc<uint32_t>(C + 64) + 32;
  uint32_t* lmc = huge_alloc<uint32_t>(V + 64) + 32;
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
      const uint32_t s = c2v[corner_next((uint32_t)c)], t = c2v[corner_prev((uint32_t)c)];
      es[c] = {((uint64_t)std::min(s, t) << 32) | std::max(s, t), (uint32_t)c};
    }
    std::sort(es.begin(), es.end(), [](const E& x, const E& y) { return x.key < y.key; });
    for (size_t c = 0; c < C; ++c) opp[c] = kNone;
    for (size_t i = 0; i + 1 < C; ++i) if (es[i].key == es[i + 1].key) { opp[es[i].c] = es[i + 1].c; opp[es[i + 1].c] = es[i].c; ++i; }
  }
  for (size_t c = C; c-- > 0;) lmc[c2v[c]] = (uint32_t)c;
  for (int r = 0; r < repeats; ++r) {
    CornerTables t;
    t.F = F; t.V = V; t.c2p = c2v; t.c2v = c2v; t.opp = opp; t.lmc = lmc; t.no_boundary = true; t.att.resize(2);
    EdgebreakerResult eb;
    std::string err;
    double t0 = now_ms();
    const int rc = run_edgebreaker(t, eb, err, nullptr);
    double t1 = now_ms();
    std::vector<uint32_t> seq;
    TableRef tr{F, V, c2v, opp, lmc};
    std::vector<uint8_t> onb(V, 0);
    attribute_sequence(tr, eb, seq, onb.data());
    double t2 = now_ms();
    std::printf("library walks on this program's tables: run_edgebreaker %.1f ms (rc %d, %zu bytes), attribute_sequence %.1f ms (%zu entries)\n", t1 - t0, rc, eb.connectivity.size(), t2 - t1, seq.size());
  }
  if (std::getenv("DMI_TRACE")) std::printf("thread time by step over %llu calls: set-up %.3f ms, traversal %.3f, bits %.3f, seam streams %.3f, whole %.3f\n", (unsigned long long)g_eb_ns[4].load(), g_eb_ns[0] / 1e6, g_eb_ns[1] / 1e6, g_eb_ns[2] / 1e6, g_eb_ns[3] / 1e6, g_eb_ns[5] / 1e6);
  if (std::FILE* f = std::fopen("/proc/self/smaps_rollup", "r")) {
    char line[256];
    while (std::fgets(line, sizeof line, f)) if (!std::strncmp(line, "Rss", 3) || !std::strncmp(line, "AnonHuge", 8)) std::fputs(line, stdout);
    std::fclose(f);
  }
  return 0;
}
