// host_chains.cpp — the serial rANS / rABS recurrence of ONE long stream on ONE host core (the hybrid form of a single large
// mesh: every data-parallel stage and the table stage stay on the GPU; the symbols and the device-built coding table come
// back over PCIe and the strict dependency chain x' = (x / f)·2^P + x % f + c runs where a dependent integer chain is
// fastest — a 5 GHz out-of-order core retires one step per ≈ 10 clocks, the gfx950 scalar unit one per ≈ 39 at 2.4 GHz).
// Batches of many meshes keep the device chains (dmi_chains.hip): thousands of streams in flight fill the chip.
// This is not a fallback: nothing here runs without the device stages before it, and a job without a HIP device fails.
//
// Reference arithmetic (draco-oxide/src/): encode/entropy/rans.rs:33-46 (RansCoder::write), :48-68 (flush), :91-108
// (RabsCoder::write), encode/entropy/symbol_coding.rs:161-163 (symbols are fed in reverse).
// The coding records are the ones k_tables / make_rans_entry build for the device walker (exact x / f by multiply-high,
// d = 2^P - f so that (x / f)·2^P + x % f = x + (x / f)·d, renormalisation threshold t = f·2^10 or f·2^12): the proof of
// exactness is in dmi_chains.hip.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dmi_device.hpp"
#include "dmi_host.hpp"
#include <memory>
#include <mutex>
#include "host_chains.hpp"

namespace dmi {

namespace {

inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

// bytes of the tagged final state (rans.rs:48-68); 0 = StateTooLarge
inline uint32_t flush_bytes(uint32_t s, uint8_t* at) {
  uint32_t nb, v;
  if (s < (1u << 6)) { nb = 1; v = s; }
  else if (s < (1u << 14)) { nb = 2; v = (1u << 14) + s; }
  else if (s < (1u << 22)) { nb = 3; v = (2u << 22) + s; }
  else if (s < (1u << 30)) { nb = 4; v = (3u << 30) + s; }
  else return 0;
  for (uint32_t k = 0; k < nb; ++k) at[k] = (uint8_t)(v >> (8 * k));
  return nb;
}

// The host form of a coding record: the device record's multiplier m and shift b folded into one 64-bit multiplier, so that the
// exact quotient is the HIGH HALF of one 64×64 multiply — x / f = (x·m) >> (32 + b) = hi64(x · (m << (32 - b))) — three clocks on
// the dependency chain instead of multiply, shift, shift.  flags: bit 8 f == 1 (x / f = x), bit 9 f < 2^(P-8) (may shed > 1 byte).
struct HostRec { uint64_t m64; uint32_t d, c, t, flags; };
inline HostRec host_rec(const RansEntry& e) {
  const uint32_t b = e.b & 31u;
  return HostRec{(uint64_t)e.m << (32u - b), e.d, e.c, e.t, e.b & 0x300u};   // (b ≤ 19: nothing is shifted out; f == 1 never multiplies)
}
inline uint32_t quot(uint32_t x, uint64_t m64) { return (uint32_t)(((unsigned __int128)x * m64) >> 64); }

// one step, any renormalisation (≤ 3 bytes: x < 2^30, t ≥ 2^10): the bytes leave low byte first, exactly the reference's loop
inline uint32_t step_general(uint32_t x, const HostRec& e, uint8_t*& p) {
  const uint64_t t = e.t;
  const uint32_t nb = (uint32_t)(x >= t) + (uint32_t)(x >= (t << 8)) + (uint32_t)(x >= (t << 16));
  std::memcpy(p, &x, 4);   // little-endian host: bytes 0..2 of x in emission order; only nb of them are kept
  p += nb;
  x >>= 8u * nb;
  const uint32_t q = (e.flags & 0x100u) ? x : quot(x, e.m64);   // f == 1: x / f = x
  return x + q * e.d + e.c;
}

}  // namespace

bool HostChainOut::reserve(size_t need) {
  if (need <= cap) return true;
  size_t want = std::max(need, cap + cap / 2 + 4096);
  uint8_t* q = static_cast<uint8_t*>(std::realloc(data, want));
  if (!q) return false;
  data = q; cap = want;
  return true;
}
HostChainOut::~HostChainOut() { std::free(data); }

// n symbols, fed last to first; table[s] = the coding record of symbol s (bins entries); state0 = 4·2^P.
// err: 0 ok, 1 StateTooLarge (rans.rs:63-65), 2 out of memory, 3 symbol outside the table.
template <class Sym>
static void rans_chain_impl(const Sym* sym, uint64_t n, const RansEntry* table, uint32_t bins, uint32_t precision, HostChainOut& out) {
  out.len = 0; out.err = 0;
  uint32_t x = 4u << precision;
  const uint32_t state0 = x;
  std::vector<HostRec> recs(bins);
  for (uint32_t k = 0; k < bins; ++k) recs[k] = host_rec(table[k]);
  const HostRec* __restrict__ rec = recs.data();
  // the output grows in steps: a block of kBlock symbols sheds at most 3·kBlock bytes (+ 4 of store slack + 4 of flush)
  constexpr uint64_t kBlock = 4096;
  if (!out.reserve(n / 2 + 3 * kBlock + 64)) { out.err = 2; return; }
  uint8_t* p = out.data;
  uint64_t i = n;
  while (i > 0) {
    const uint64_t cnt = std::min<uint64_t>(kBlock, i);
    {
      const size_t used = (size_t)(p - out.data);
      if (used + 3 * cnt + 16 > out.cap) { if (!out.reserve(used + used / 2 + 3 * cnt + 64)) { out.err = 2; return; } p = out.data + used; }
    }
    const uint64_t stop = i - cnt;
    for (; i > stop; --i) {
      const uint32_t s = (uint32_t)sym[i - 1];
      if (__builtin_expect(s >= bins, 0)) { out.err = 3; return; }
      const HostRec e = rec[s];
      if (__builtin_expect(e.flags == 0u, 1)) {
        // f ≥ 2^(P-8): the state can shed at most one byte (x < 2^(P+10) ≤ t·2^8); branch-free — the byte leaves about every other step
        *p = (uint8_t)x;
        const bool r = x >= e.t;
        p += r;
        const uint32_t xs = __builtin_unpredictable(r) ? (x >> 8) : x;
        x = (xs + e.c) + quot(xs, e.m64) * e.d;
      } else {
        if (__builtin_expect(e.t == 0u, 0)) { out.err = 3; return; }   // a symbol the table does not code (never after a clean histogram)
        x = step_general(x, e, p);
      }
    }
  }
  const uint32_t nb = flush_bytes(x - state0, p);
  if (!nb) { out.err = 1; return; }
  p += nb;
  out.len = (size_t)(p - out.data);
}

void host_rans_chain(const uint32_t* sym, uint64_t n, const RansEntry* table, uint32_t bins, uint32_t precision, HostChainOut& out) {
  rans_chain_impl(sym, n, table, bins, precision, out);
}
void host_rans_chain16(const uint16_t* sym, uint64_t n, const RansEntry* table, uint32_t bins, uint32_t precision, HostChainOut& out) {
  rans_chain_impl(sym, n, table, bins, precision, out);
}

// n bits, fed first to last (mesh_normal_prediction.rs:154-157, mesh_prediction_for_texture_coordinates.rs:241-256);
// e[0] / e[1] = coding records of bit 0 / bit 1 (make_rans_entry(p0, 256 - p0, 8) / make_rans_entry(256 - p0, 0, 8)).
void host_rabs_chain(const uint8_t* bits, uint64_t n, const RansEntry* e, HostChainOut& out) {
  out.len = 0; out.err = 0;
  uint32_t x = 4096u;
  if (!out.reserve(n + 64)) { out.err = 2; return; }   // ≤ 1 byte per bit (single `if`, rans.rs:97)
  uint8_t* p = out.data;
  const HostRec h[2] = {host_rec(e[0]), host_rec(e[1])};
  const bool flagged = ((h[0].flags | h[1].flags) & 0x100u) != 0u;   // p0 ∈ {1, 255}: one of the frequencies is 1
  if (!flagged) {
    // both records side by side so that the select is an index, not a branch
    const uint64_t m[2] = {h[0].m64, h[1].m64};
    const uint32_t d[2] = {h[0].d, h[1].d}, c[2] = {h[0].c, h[1].c}, t[2] = {h[0].t, h[1].t};
    for (uint64_t i = 0; i < n; ++i) {
      const uint32_t k = bits[i] != 0;
      *p = (uint8_t)x;
      const bool r = x >= t[k];
      p += r;
      const uint32_t xs = __builtin_unpredictable(r) ? (x >> 8) : x;
      x = (xs + c[k]) + quot(xs, m[k]) * d[k];
    }
  } else {
    for (uint64_t i = 0; i < n; ++i) {
      const HostRec& r = h[bits[i] != 0];
      if (x >= r.t) { *p++ = (uint8_t)x; x >>= 8; }
      const uint32_t q = (r.flags & 0x100u) ? x : quot(x, r.m64);
      x = x + q * r.d + r.c;
    }
  }
  const uint32_t nb = flush_bytes(x - 4096u, p);
  if (!nb) { out.err = 1; return; }
  p += nb;
  out.len = (size_t)(p - out.data);
}

bool host_rabs_bytes(uint8_t zero_prob, const uint8_t* fed, uint64_t n, std::vector<uint8_t>& bytes) {
  const uint32_t p0 = zero_prob, f1 = 256u - p0;
  const RansEntry e[2] = {make_rans_entry(p0, f1, 8), make_rans_entry(f1, 0, 8)};
  HostChainOut o;
  host_rabs_chain(fed, n, e, o);
  if (o.err) return false;
  bytes.assign(o.data, o.data + o.len);
  return true;
}

// n copies of ONE bit (the seam-flag stream of an attribute without seams: 1.5 flags per face, all zero).  With the bit fixed a step is a
// function of the state alone, so the states — and with them the bytes that leave — repeat: for zero_prob 255 after 1410 steps with a
// period of 1409 steps and one byte.  The first window of steps runs as the coder runs them, the period is read off the recorded states, the
// rest of the stream is the period's bytes over and over; a window without a repeat falls back to the plain loop.  Same bytes as
// host_rabs_bytes on n equal bits (tests/test_host_chains.py).
// The window (states, byte offsets, bytes, period) depends on (zero_prob, bit) alone — and the stream every seam-free attribute of every mesh gets is
// the same one (all zeros at zero_prob 255) — so it is computed once per process and shared: a call then copies its bytes (a batch of 256 meshes
// spent 21 ms of thread time re-running the 16 K-step window per mesh).
namespace {
struct ConstantWindow {
  std::vector<uint32_t> xs, ob;   // state before step i, bytes written before step i (i = 0 … W)
  std::vector<uint8_t> bytes;     // the bytes of the first W steps
  uint64_t period = 0;            // 0: no repeat inside the window
  HostRec h;
};
inline uint32_t constant_step(const HostRec& h, uint32_t x, std::vector<uint8_t>& out) {   // host_rabs_chain's generic loop body
  if (x >= h.t) { out.push_back((uint8_t)x); x >>= 8; }
  const uint32_t q = (h.flags & 0x100u) ? x : quot(x, h.m64);
  return x + q * h.d + h.c;
}
constexpr uint64_t kConstantWindow = 1u << 14;
std::shared_ptr<const ConstantWindow> constant_window(uint8_t zero_prob, uint32_t bit) {
  static std::mutex m;
  static std::shared_ptr<const ConstantWindow> cache[512];
  const size_t key = (size_t)zero_prob | (bit ? 256u : 0u);
  {
    std::lock_guard<std::mutex> lock(m);
    if (cache[key]) return cache[key];
  }
  auto w = std::make_shared<ConstantWindow>();
  const uint32_t p0 = zero_prob, f1 = 256u - p0;
  w->h = host_rec(bit ? make_rans_entry(f1, 0, 8) : make_rans_entry(p0, f1, 8));
  const uint64_t W = kConstantWindow;
  w->xs.resize((size_t)W + 1); w->ob.resize((size_t)W + 1);
  uint32_t x = 4096u;
  for (uint64_t i = 0; i < W; ++i) { w->xs[(size_t)i] = x; w->ob[(size_t)i] = (uint32_t)w->bytes.size(); x = constant_step(w->h, x, w->bytes); }
  w->xs[(size_t)W] = x; w->ob[(size_t)W] = (uint32_t)w->bytes.size();
  for (uint64_t l = 1; l <= W; ++l) if (w->xs[(size_t)(W - l)] == x) { w->period = l; break; }
  std::lock_guard<std::mutex> lock(m);
  if (!cache[key]) cache[key] = w;
  return cache[key];
}
}  // namespace
bool host_rabs_constant(uint8_t zero_prob, uint32_t bit, uint64_t n, std::vector<uint8_t>& bytes) {
  const std::shared_ptr<const ConstantWindow> cw = constant_window(zero_prob, bit);
  const ConstantWindow& w = *cw;
  const uint64_t W = std::min<uint64_t>(n, kConstantWindow);
  bytes.assign(w.bytes.begin(), w.bytes.begin() + (long)w.ob[(size_t)W]);
  uint32_t x = w.xs[(size_t)W];
  uint64_t left = n - W;
  if (left) {   // (W = the whole window here)
    const uint64_t period = w.period;
    if (period) {
      const size_t from = w.ob[(size_t)(W - period)], per_bytes = w.ob[(size_t)W] - from;
      const uint64_t reps = left / period, rem = left % period;
      const uint8_t* cycle = w.bytes.data() + from;
      bytes.reserve(bytes.size() + (size_t)reps * per_bytes + per_bytes + 8);
      if (per_bytes == 1) bytes.insert(bytes.end(), (size_t)reps, cycle[0]);
      else for (uint64_t r = 0; r < reps; ++r) bytes.insert(bytes.end(), cycle, cycle + per_bytes);
      const size_t tail = w.ob[(size_t)(W - period + rem)] - from;
      bytes.insert(bytes.end(), cycle, cycle + tail);
      x = w.xs[(size_t)(W - period + rem)];
    } else {
      for (uint64_t i = 0; i < left; ++i) x = constant_step(w.h, x, bytes);
    }
  }
  uint8_t fl[4];
  const uint32_t nb = flush_bytes(x - 4096u, fl);
  if (!nb) return false;
  bytes.insert(bytes.end(), fl, fl + nb);
  return true;
}

// ---- the inverse coders (decode/entropy/rans.rs:36-69, :106-127): the decoder side's serial stage, on a host core ----
namespace {
inline bool read_tagged_state(const uint8_t* data, size_t& pos, uint64_t& state) {   // rans.rs:36-46: the last byte's top two bits give the state's width
  if (pos == 0) return false;
  const uint8_t meta = data[--pos];
  const uint32_t flag = meta >> 6;
  if (pos < flag) return false;
  uint64_t s = 0;
  for (uint32_t k = 0; k < flag; ++k) s |= (uint64_t)data[pos - flag + k] << (8 * k);
  pos -= flag;
  state = s | ((uint64_t)(meta & 0x3F) << (flag << 3));
  return true;
}
}  // namespace

// n symbols in FORWARD order (the encoder fed them last to first); freq[] = normalised frequencies summing to 2^precision.
// Returns false on a truncated stream / a frequency table that does not sum to 2^precision.
bool host_rans_decode(const uint8_t* data, size_t len, const uint32_t* freq, uint32_t num_symbols, uint32_t precision, uint64_t n, uint32_t* out) {
  size_t pos = len;
  uint64_t x;
  if (!read_tagged_state(data, pos, x)) return false;
  const uint64_t L = (uint64_t)4 << precision, mask = ((uint64_t)1 << precision) - 1;
  x += L;
  std::vector<uint32_t> slot((size_t)1 << precision), cum(num_symbols);
  uint64_t c = 0;
  for (uint32_t s = 0; s < num_symbols; ++s) {
    cum[s] = (uint32_t)c;
    if (c + freq[s] > ((uint64_t)1 << precision)) return false;
    for (uint32_t k = 0; k < freq[s]; ++k) slot[c + k] = s;
    c += freq[s];
  }
  if (c != ((uint64_t)1 << precision)) return false;
  for (uint64_t i = 0; i < n; ++i) {
    while (x < L) { if (pos == 0) return false; x = x * 256 + data[--pos]; }
    const uint64_t q = x >> precision, r = x & mask;
    const uint32_t s = slot[r];
    x = q * freq[s] + r - cum[s];
    out[i] = s;
  }
  return true;
}
// n bits in the order the decoder pops them (= the REVERSE of the order the encoder pushed them)
bool host_rabs_decode(const uint8_t* data, size_t len, uint32_t zero_prob, uint64_t n, uint8_t* out) {
  size_t pos = len;
  uint64_t x;
  if (!read_tagged_state(data, pos, x)) return false;
  x += 4096;
  const uint64_t f1 = 256 - zero_prob;
  for (uint64_t i = 0; i < n; ++i) {
    if (x < 4096) { if (pos == 0) return false; x = (x << 8) + data[--pos]; }
    const uint64_t q = x >> 8, r = x & 255, xn = q * f1;
    if (r < f1) { x = xn + r; out[i] = 1; } else { x = x - xn - f1; out[i] = 0; }
  }
  return true;
}

}  // namespace dmi
