// oracle/orc_entropy.cpp — TEST INFRASTRUCTURE (see oracle.hpp header).
// Restatement of encode/entropy/{rans,symbol_coding}.rs and shared/entropy/mod.rs, plus the inverse
// (decode/entropy/rans.rs) used only for round-trip self-checks.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <numeric>

#include "oracle.hpp"

namespace orc {

// RansCoder::new (rans.rs:18-31) + rans_build_tables (shared/entropy/mod.rs:41-64)
std::string RansCoder::init(const std::vector<u64>& dist, u32 precision_bits) {
  precision = precision_bits;
  l_base = (1ull << precision) << 2;
  state = l_base;
  freq.resize(dist.size());
  cum.resize(dist.size());
  u64 c = 0;
  for (size_t i = 0; i < dist.size(); ++i) { freq[i] = (u32)dist[i]; cum[i] = (u32)c; c += dist[i]; }
  if (c != (1ull << precision)) return "FrequencyCountNotCompatibleWithRansPrecision";
  out.clear();
  return "";
}

// rans.rs:33-46
std::string RansCoder::write(u64 idx) {
  if (idx >= freq.size()) return "InvalidSymbolIndex";
  u64 f = freq[idx];
  if (f == 0) return "symbol with zero normalised frequency (the reference loops forever / divides by zero here, rans.rs:40-44)";
  // Rust precedence: `*` before `<<`:  ((l_base >> P) * f) << 8
  while (state >= (((l_base >> precision) * f) << 8)) {
    out.w8((u8)(state & 0xFF));
    state >>= 8;
  }
  // f == 0 would be a division-by-zero panic in the reference.
  if (f == 0) return "division by zero (symbol with zero normalised frequency)";
  state = ((state / f) << precision) + state % f + cum[idx];
  return "";
}

static std::string flush_state(u64 state, Bytes& out) {   // rans.rs:48-68 / :110-128
  if (state < (1ull << 6)) out.w8((u8)((0x00u << 6) + (u8)state));
  else if (state < (1ull << 14)) out.w16((u16)((0x01u << 14) + (u16)state));
  else if (state < (1ull << 22)) out.w24((u32)((0x02u << 22) + (u32)state));
  else if (state < (1ull << 30)) out.w32((u32)((0x03u << 30) + (u32)state));
  else return "StateTooLarge";
  return "";
}

std::string RansCoder::flush(Bytes& dst) {
  state -= l_base;
  std::string e = flush_state(state, out);
  if (!e.empty()) return e;
  dst = out;
  return "";
}

// rans.rs:91-108 — single `if` renormalisation (quirk Q21), precision 8, L = 4096
void RabsCoder::write(u8 value) {
  const u64 P = 8, L = 4096;
  u64 f1 = (1ull << P) - p0;
  u64 f = value > 0 ? f1 : p0;
  if (state >= (((L >> P) * f) << 8)) { out.w8((u8)(state & 0xFF)); state >>= 8; }
  u64 q = state / f, r = state % f;
  state = (q << P) + r + (value > 0 ? 0 : f1);
}
std::string RabsCoder::flush(Bytes& dst) {
  state -= 4096;
  std::string e = flush_state(state, out);
  if (!e.empty()) return e;
  dst = out;
  return "";
}

// RansSymbolEncoder::new, rans.rs:146-239: normalise to 2^P and serialise the table.
static std::string normalise_and_write_table(const std::vector<u64>& freq_counts, u32 P, Bytes& w, std::vector<u64>& distribution) {
  double total_freq = 0.0;
  {
    u64 s = 0;
    for (u64 f : freq_counts) s += f;
    total_freq = (double)s;   // sum::<usize>() as f64
  }
  size_t num_symbols = 0;
  for (size_t i = freq_counts.size(); i-- > 0;) if (freq_counts[i] > 0) { num_symbols = i + 1; break; }
  if (num_symbols == 0) return "empty histogram (reference unwrap() panic, rans.rs:152)";
  const u64 rans_precision = 1ull << P;
  distribution.assign(num_symbols, 0);
  u64 total_rans_prob = 0;
  for (size_t i = 0; i < num_symbols; ++i) {
    u64 freq = freq_counts[i];
    double prob = (double)freq / total_freq;
    u64 new_freq = (u64)(prob * (double)rans_precision + 0.5);
    if (new_freq == 0 && freq > 0) new_freq = 1;
    distribution[i] = new_freq;
    total_rans_prob += new_freq;
  }
  if (std::getenv("ORC_TRACE")) std::fprintf(stderr, "[orc] normalise: num_symbols %zu total %llu target %llu total_freq %.1f\n", num_symbols, (unsigned long long)total_rans_prob, (unsigned long long)rans_precision, total_freq);
  if (total_rans_prob != rans_precision) {
    std::vector<size_t> sorted(num_symbols);
    std::iota(sorted.begin(), sorted.end(), 0);
    std::stable_sort(sorted.begin(), sorted.end(), [&](size_t a, size_t b) { return distribution[a] < distribution[b]; });   // sort_by_key is stable (Q12)
    if (total_rans_prob < rans_precision) {
      distribution[sorted.back()] += rans_precision - total_rans_prob;
    } else {
      u64 err = total_rans_prob - rans_precision;
      size_t i = distribution.size() - 1;
      while (err > 0) {
        if (distribution[sorted[i]] == 0) return "normalisation underflow (reference would panic)";
        distribution[sorted[i]] -= 1;
        if (i == 0 && err > 1) return "normalisation index underflow (reference would panic)";
        i -= 1;
        err -= 1;
      }
    }
  }
  // serialise :195-231
  leb128_write(num_symbols, w);
  size_t i = 0;
  while (i < num_symbols) {
    u64 freq = distribution[i];
    if (freq == 0) {
      size_t offset = 0;
      while (offset < (1u << 6)) {
        if (i + offset + 1 >= distribution.size()) return "zero-run scan out of bounds (reference would panic)";
        u64 next_prob = distribution[i + offset + 1];
        if (next_prob > 0) { i += offset; break; }
        offset += 1;
      }
      w.w8((u8)((((u8)offset) << 2) | 3));   // Q20
    } else {
      u32 extra = 0;
      if (freq >= (1u << 6)) { extra += 1; if (freq >= (1u << 14)) { extra += 1; if (freq >= (1u << 22)) return "RANS precision too high"; } }
      w.w8((u8)((freq << 2) | (extra & 3)));
      for (u32 b = 0; b < extra; ++b) w.w8((u8)(freq >> (8 * (b + 1) - 2)));
    }
    i += 1;
  }
  return "";
}

static u32 precision_for_bit_length(u32 bit_length) {   // symbol_coding.rs:120-140
  static const u32 tab[19] = {0, 12, 12, 12, 12, 12, 12, 12, 12, 13, 15, 16, 18, 19, 20, 20, 20, 20, 20};
  return tab[bit_length];
}

// encode_symbols(.., DirectCoded) symbol_coding.rs:17-55,109-166
std::string encode_symbols_direct(const std::vector<u32>& symbols, Bytes& w) {
  w.w8(1);   // SymbolEncodingMethod::DirectCoded.write_to, shared/entropy/mod.rs:33-36
  std::unique_ptr<StageTimer> t_tab(new StageTimer(7));
  u64 num_nonzero = 0;
  for (u32 s : symbols) if (s > 0) ++num_nonzero;   // :46 (quirk Q11)
  u32 bl = (u32)(64 - (num_nonzero ? __builtin_clzll(num_nonzero) : 64)) + 1;
  if (bl < 1) bl = 1;
  if (bl > 18) bl = 18;
  w.w8((u8)bl);
  const u32 P = precision_for_bit_length(bl);
  // histogram :149-157
  std::vector<u64> freq_counts;
  u64 max_symbol = 0;
  for (u32 s : symbols) {
    if (s >= max_symbol) { max_symbol = s; freq_counts.resize(max_symbol + 1, 0); }
    freq_counts[s] += 1;
  }
  std::vector<u64> dist;
  std::string e = normalise_and_write_table(freq_counts, P, w, dist);
  if (!e.empty()) return e;
  RansCoder rc;
  e = rc.init(dist, P);
  if (!e.empty()) return e;
  t_tab.reset();
  StageTimer t_rans(8), t_rans_only(9);
  g_rans_symbols += (double)symbols.size();
  for (size_t i = symbols.size(); i-- > 0;) {   // :161-163 reversed
    e = rc.write(symbols[i]);
    if (!e.empty()) return e;
  }
  Bytes b;
  e = rc.flush(b);
  if (!e.empty()) return e;
  leb128_write(b.size(), w);   // rans.rs:248-255
  w.append(b);
  return "";
}

// ------------------------------- inverse (self-check only) -----------------------------------
static bool read_tagged_state(const u8* data, size_t& pos, u64& state) {   // decode/entropy/rans.rs:36-46
  if (pos == 0) return false;
  u8 meta = data[--pos];
  u32 flag = meta >> 6;
  u64 s = 0;
  if (pos < flag) return false;
  // read_uN_back reads the preceding bytes as a little-endian integer
  for (u32 k = 0; k < flag; ++k) s |= (u64)data[pos - flag + k] << (8 * k);
  pos -= flag;
  s |= (u64)(meta & 0x3F) << (flag << 3);
  state = s;
  return true;
}

std::string rans_decode_stream(const u8* data, size_t len, const std::vector<u64>& dist, u32 P, size_t n, std::vector<u32>& out) {
  size_t pos = len;
  u64 state;
  if (!read_tagged_state(data, pos, state)) return "NotEnoughData";
  const u64 L = (1ull << P) << 2;
  state += L;
  std::vector<u32> slot(1ull << P);
  std::vector<u64> cum(dist.size());
  u64 c = 0;
  for (size_t i = 0; i < dist.size(); ++i) { cum[i] = c; for (u64 k = 0; k < dist[i]; ++k) slot[c + k] = (u32)i; c += dist[i]; }
  if (c != (1ull << P)) return "bad table";
  out.resize(n);
  for (size_t i = 0; i < n; ++i) {   // decode/entropy/rans.rs:58-69
    while (state < L) { if (pos == 0) return "NotEnoughData"; state = state * 256 + data[--pos]; }
    u64 q = state >> P, r = state & ((1ull << P) - 1);
    u32 s = slot[r];
    state = q * dist[s] + r - cum[s];
    out[i] = s;
  }
  // (bytes may remain: when the FIRST symbol the encoder coded is rare, its renormalisation sheds bytes of the initial state, which no
  //  decoder reads back — the reference's reverse reader simply stops, decode/entropy/rans.rs:58-69)
  return "";
}

std::string rabs_decode_stream(const u8* data, size_t len, u64 p0, size_t n, std::vector<u8>& out) {
  size_t pos = len;
  u64 state;
  if (!read_tagged_state(data, pos, state)) return "NotEnoughData";
  state += 4096;
  const u64 f1 = 256 - p0;
  out.resize(n);
  for (size_t i = 0; i < n; ++i) {   // decode/entropy/rans.rs:106-127
    if (state < 4096) { if (pos == 0) return "NotEnoughData"; state = (state << 8) + data[--pos]; }
    u64 x = state, q = x >> 8, r = x & 255, xn = q * f1;
    if (r < f1) { state = xn + r; out[i] = 1; } else { state = x - xn - f1; out[i] = 0; }
  }
  return "";   // (see rans_decode_stream: unread bytes of the initial state are legal)
}

// decode/entropy/symbol_coding.rs (direct-coded branch) + RansSymbolDecoder::new rans.rs:139-200.
std::string decode_symbols_direct(const u8* data, size_t len, size_t n, std::vector<u32>& out, size_t* consumed) {
  size_t p = 0;
  auto rd = [&](u8& b) { if (p >= len) return false; b = data[p++]; return true; };
  auto leb = [&](u64& v) { v = 0; u32 sh = 0; u8 b; do { if (!rd(b)) return false; v |= (u64)(b & 0x7F) << sh; sh += 7; } while (b & 0x80); return true; };
  u8 method, bl;
  if (!rd(method) || method != 1) return "not direct coded";
  if (!rd(bl) || bl < 1 || bl > 18) return "bad bit length";
  const u32 P = precision_for_bit_length(bl);
  u64 num_symbols;
  if (!leb(num_symbols)) return "eof";
  std::vector<u64> dist(num_symbols, 0);
  size_t i = 0;
  while (i < num_symbols) {
    u8 b;
    if (!rd(b)) return "eof";
    u32 token = b & 3;
    if (token == 3) {
      u32 offset = b >> 2;
      if (i + offset >= num_symbols) return "Invalid offset for frequency counts";
      i += offset;
    } else {
      u64 count = b >> 2;
      for (u32 j = 0; j < token; ++j) { u8 eb; if (!rd(eb)) return "eof"; count |= (u64)eb << (8 * (j + 1) - 2); }
      dist[i] = count;
    }
    i += 1;
  }
  u64 nbytes;
  if (!leb(nbytes)) return "eof";
  if (p + nbytes > len) return "eof";
  std::vector<u32> rev;
  std::string e = rans_decode_stream(data + p, nbytes, dist, P, n, rev);
  if (!e.empty()) return e;
  // the encoder fed symbols in reverse, so the decoder yields them in forward order
  out = rev;
  p += nbytes;
  if (consumed) *consumed = p;
  return "";
}

}  // namespace orc
