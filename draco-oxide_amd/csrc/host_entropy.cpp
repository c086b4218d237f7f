// host_entropy.cpp — frequency-table normalisation + serialisation on the host (the host form of the table stage: DMI_HOST_TABLES=1
// and ToBits attributes; the device form is k_tables in dmi_chains.hip — tests compare the two).
// Reproduces RansSymbolEncoder::new (encode/entropy/rans.rs:146-239) and the DirectCoded prologue
// of encode_symbols (encode/entropy/symbol_coding.rs:17-55,109-141) from a symbol histogram.
#include <algorithm>
#include <numeric>

#include "dmi_host.hpp"

namespace dmi {

int FreqTable::build(const uint32_t* hist, size_t bins, std::string& err) {
  // number of non-zero symbols decides bit_length, which decides the precision (quirk Q11)
  uint64_t total = 0;
  size_t num_symbols = 0;
  for (size_t s = 0; s < bins; ++s) { total += hist[s]; if (hist[s]) num_symbols = s + 1; }
  if (num_symbols == 0) { err = "empty symbol histogram"; return DMI_ERR_ENTROPY; }
  const uint64_t nonzero = total - hist[0];
  unsigned bl = 1;
  for (uint64_t x = nonzero; x; x >>= 1) ++bl;          // (64 - leading_zeros) + 1
  bit_length = std::min(18u, std::max(1u, bl));
  static const uint8_t prec_of[19] = {0, 12, 12, 12, 12, 12, 12, 12, 12, 13, 15, 16, 18, 19, 20, 20, 20, 20, 20};
  precision = prec_of[bit_length];
  const uint64_t target = 1ull << precision;

  freq.assign(num_symbols, 0);
  uint64_t sum = 0;
  const double tot = (double)total;
  for (size_t s = 0; s < num_symbols; ++s) {
    uint64_t nf = (uint64_t)(((double)hist[s] / tot) * (double)target + 0.5);
    if (nf == 0 && hist[s]) nf = 1;
    freq[s] = (uint32_t)nf;
    sum += nf;
  }
  if (sum != target) {
    // order by (normalised frequency, index): the reference's stable sort_by_key (Q12)
    if (sum < target) {
      size_t best = 0;
      for (size_t s = 1; s < num_symbols; ++s) if (freq[s] >= freq[best]) best = s;   // last of the largest
      freq[best] += (uint32_t)(target - sum);
    } else {
      uint64_t excess = sum - target;
      if (excess > num_symbols) { err = "frequency normalisation overflow"; return DMI_ERR_ENTROPY; }
      std::vector<uint32_t> order(num_symbols);
      std::iota(order.begin(), order.end(), 0u);
      // only the `excess` largest are needed, largest first, ties: higher index first
      auto greater = [&](uint32_t a, uint32_t b) { return freq[a] != freq[b] ? freq[a] > freq[b] : a > b; };
      if (excess < num_symbols) std::partial_sort(order.begin(), order.begin() + excess, order.end(), greater);
      else std::sort(order.begin(), order.end(), greater);
      for (uint64_t k = 0; k < excess; ++k) {
        if (freq[order[k]] == 0) { err = "frequency normalisation underflow"; return DMI_ERR_ENTROPY; }
        --freq[order[k]];
      }
    }
  }
  // The reference's over-correction can drive an occurring symbol to frequency 0; its coder then never
  // terminates (`while state >= 0`, rans.rs:40) — there is no reference output for such a histogram.
  for (size_t s = 0; s < num_symbols; ++s)
    if (hist[s] && !freq[s]) { err = "normalised frequency of an occurring symbol is zero (the reference encoder does not terminate on this input)"; return DMI_ERR_ENTROPY; }
  cum.resize(num_symbols);
  uint32_t c = 0;
  for (size_t s = 0; s < num_symbols; ++s) { cum[s] = c; c += freq[s]; }

  ByteSink h;
  h.u8(1);                       // SymbolEncodingMethod::DirectCoded
  h.u8((uint8_t)bit_length);
  h.leb128(num_symbols);
  for (size_t i = 0; i < num_symbols; ++i) {
    const uint32_t f = freq[i];
    if (f == 0) {
      // zero-run token: (number of further zeros, ≤63) << 2 | 3   (Q20)
      size_t run = 0;
      bool hit = false;
      while (run < 64) {
        if (i + run + 1 >= num_symbols) { err = "zero-run past the table end"; return DMI_ERR_ENTROPY; }
        if (freq[i + run + 1] > 0) { hit = true; break; }
        ++run;
      }
      h.u8((uint8_t)((uint8_t)run << 2) | 3);
      if (hit) i += run;
    } else {
      const unsigned extra = f >= (1u << 14) ? 2 : (f >= (1u << 6) ? 1 : 0);
      if (f >= (1u << 22)) { err = "frequency too large"; return DMI_ERR_ENTROPY; }
      h.u8((uint8_t)((f << 2) | extra));
      for (unsigned b = 0; b < extra; ++b) h.u8((uint8_t)(f >> (8 * (b + 1) - 2)));
    }
  }
  header.swap(h.b);
  return DMI_OK;
}

}  // namespace dmi
