// dmi_chains.hip — the serial entropy coders (encode/entropy/rans.rs:33-68 rANS, :91-128 rABS) for gfx950.
//
// The rANS/rABS state recurrence is ONE dependency chain per stream (a single non-interleaved
// stream per attribute), so a stream is owned by one wavefront and the chain runs on the SCALAR unit:
//   * prep kernels (data-parallel, all CUs) turn the stream into 20-byte coding records {m, b, d, c, t}
//     in coding order, so the chain never touches a symbol table:
//       k_rans_prep      symbols, reversed (symbol_coding.rs:161-163)
//       k_bits_prep      rABS bits, forward (normal flips, mesh_normal_prediction.rs:154-157)
//       k_orient_prep    orientation flags → compacted transition bits, forward
//                        (mesh_prediction_for_texture_coordinates.rs:241-256)
//   * k_chains: per 64 records a walker wavefront walks 64 steps on SGPRs (records arrive through s_load_dwordx16,
//     one chunk of 8 ahead) — per step: exact x/f by multiply-high, renormalisation shift, state update — parking
//     pre-renormalisation states in lanes; its emitter wavefront turns them into bytes at prefix-sum offsets.
//
// rABS is the same recurrence with precision 8 and L = 4096 instead of 4·2^P (rans.rs:78-108): the
// renormalisation threshold is f·2^12 instead of f·2^10, nothing else changes, so both run the same loop.
//
// Exact division (x < 2^30, 1 ≤ f ≤ 2^20):  e = floor(log2 f)
//     f not a power of two:  m = ceil(2^(32+e) / f) ∈ (2^31, 2^32),  x / f = mulhi(x, m) >> e
//        (m·f = 2^(32+e) + ε with ε < f < 2^(e+1), and x·ε < 2^(31+e) < 2^(32+e) ⇒ exact)
//     f = 2^e, e ≥ 1:        m = 2^31,                                x / f = mulhi(x, m) >> (e-1)
//     f = 1:                 flagged (bit 8 of b); the batch takes the generic (divide) loop.
// Renormalisation: `while x ≥ f·2^T { x >>= 8 }`  ⇔  bytes = (bitlen(x/f) - (T-7)) >> 3      (T = 10 / 12)
//     (x ≥ f·2^m ⇔ floor(x/f) ≥ 2^m, and floor(floor(x/2^8)/f) = floor(x/f) >> 8).
#include "dmi_device.hpp"
#include <cstring>

namespace dmi {
namespace {

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {   // popcount of mask bits below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t flush_state(uint32_t s, uint8_t* out, uint64_t pos, uint64_t cap, uint32_t& err) {   // rans.rs:48-68
  uint32_t nb, v;
  if (s < (1u << 6)) { nb = 1; v = s; }
  else if (s < (1u << 14)) { nb = 2; v = (1u << 14) + s; }
  else if (s < (1u << 22)) { nb = 3; v = (2u << 22) + s; }
  else if (s < (1u << 30)) { nb = 4; v = (3u << 30) + s; }
  else { err = 1; return 0; }
  if (pos + nb > cap) { err = 2; return 0; }
  for (uint32_t k = 0; k < nb; ++k) out[pos + k] = (uint8_t)(v >> (8 * k));
  return nb;
}

// select between two coding records field by field (a struct-valued ?: goes through private memory)
__device__ __forceinline__ RansEntry pick(bool one, const RansEntry& e0, const RansEntry& e1) {
  return RansEntry{one ? e1.m : e0.m, one ? e1.b : e0.b, one ? e1.d : e0.d, one ? e1.c : e0.c, one ? e1.t : e0.t};
}

// ---- prep kernels ---------------------------------------------------------------------------------
// Also writes one flag per batch of 64 records: "contains a frequency-1 symbol" (each wavefront covers one
// aligned batch: the grid stride is a multiple of 256).
__device__ __forceinline__ void k_rans_prep_body(const RansPrepArgs& a, const uint32_t blk_, const uint32_t nblk_) {
  const void* __restrict__ sym = a.sym;
  const bool s16 = a.sym16 != 0u;
  const uint64_t n = a.n;
  const RansEntry* __restrict__ table = a.table;
  RansEntry* __restrict__ rec = a.rec;
  uint32_t* __restrict__ batch_flags = a.batch_flags;
  const uint64_t n_round = (n + 63) & ~(uint64_t)63;
  for (uint64_t t = (uint64_t)blk_ * 256 + threadIdx.x; t < n_round; t += (uint64_t)nblk_ * 256) {
    RansEntry e{0u, 0u, 0u, 0u, 0u};
    if (t < n) { const uint32_t v = s16 ? (uint32_t)static_cast<const uint16_t*>(sym)[n - 1 - t] : static_cast<const uint32_t*>(sym)[n - 1 - t]; if (v < a.bins) e = table[v]; rec[t] = e; }
    const unsigned long long f1 = __ballot((e.b & 0x100u) != 0), multi = __ballot((e.b & 0x200u) != 0);
    if ((threadIdx.x & 63) == 0) batch_flags[t >> 6] = (f1 != 0ull ? 1u : 0u) | (multi != 0ull ? 2u : 0u);
  }
}

// Per-batch "contains a frequency-1 record" flags for a finished record stream (rABS streams).
__device__ __forceinline__ void k_batch_flags_body(const BatchFlagsArgs& a, const uint32_t blk_, const uint32_t nblk_) {
  const RansEntry* __restrict__ rec = a.rec;
  const uint64_t n = a.n_dev ? min((uint64_t)*a.n_dev, a.n) : a.n;   // (the grid is sized for a.n)
  uint32_t* __restrict__ batch_flags = a.batch_flags;
  const uint64_t n_round = (n + 63) & ~(uint64_t)63;
  for (uint64_t t = (uint64_t)blk_ * 256 + threadIdx.x; t < n_round; t += (uint64_t)nblk_ * 256) {
    const uint32_t b = (t < n) ? rec[t].b : 0u;
    const unsigned long long f1 = __ballot((b & 0x100u) != 0), multi = __ballot((b & 0x200u) != 0);
    if ((threadIdx.x & 63) == 0) batch_flags[t >> 6] = (f1 != 0ull ? 1u : 0u) | (multi != 0ull ? 2u : 0u);
  }
}

__device__ __forceinline__ void k_bits_prep_body(const BitsPrepArgs& a, const uint32_t blk_, const uint32_t nblk_) {
  const uint8_t* __restrict__ bits = a.bits;
  const uint64_t n = a.n;
  const RansEntry e0 = a.entries ? a.entries[0] : a.e0, e1 = a.entries ? a.entries[1] : a.e1;
  RansEntry* __restrict__ rec = a.rec;
  for (uint64_t t = (uint64_t)blk_ * 256 + threadIdx.x; t < n; t += (uint64_t)nblk_ * 256) rec[t] = pick(bits[t] != 0, e0, e1);
}

// One wavefront per chunk of 4096 orientation flags {0 none, 1 false, 2 true}.  The coded bit of valid
// entry j is (o[j] == o[j+1]) where o[j+1] is the next valid entry, or `true` after the last one.
// chunk_info[c] = {compact offset of the chunk, value of the first valid entry after the chunk (1 if none)}.
__device__ __forceinline__ void k_orient_prep_body(const OrientPrepArgs& a, const uint32_t blk_, const uint32_t nblk_) {
  const uint8_t* __restrict__ orient = a.orient;
  const uint32_t n = a.n;
  const uint32_t* __restrict__ chunk_info = a.chunk_info;
  const RansEntry e0 = a.entries ? a.entries[0] : a.e0, e1 = a.entries ? a.entries[1] : a.e1;
  RansEntry* __restrict__ rec = a.rec;
  const uint32_t lane = threadIdx.x;
  const uint32_t lo = blk_ * 4096u;
  const uint32_t nb = (min(n, lo + 4096u) - lo + 63u) / 64u;   // batches in this chunk (≤ 64)
  // the chunk goes through LDS with 16-byte loads (as in the summary kernel): a wavefront owns a whole chunk, a launch has about one
  // wavefront per SIMD, and 2 × 64 global byte loads waited for one after the other were 39 µs of latency per launch
  __shared__ __attribute__((aligned(16))) uint8_t staged[4096];
#pragma unroll
  for (uint32_t w = 0; w < 4; ++w) {
    const uint32_t off = (w * 64u + lane) * 16u, at = lo + off;
    if (at + 16u <= n) *reinterpret_cast<uint4*>(staged + off) = *reinterpret_cast<const uint4*>(orient + at);
    else for (uint32_t k = 0; k < 16u; ++k) staged[off + k] = (at + k < n) ? orient[at + k] : (uint8_t)0;
  }
  __syncthreads();
  // pass 1: valid count per batch → exclusive offsets (lane b owns batch b)
  uint32_t cnt = 0;
  for (uint32_t b = 0; b < nb; ++b) {
    const uint32_t f = staged[b * 64u + lane];
    const uint32_t c = (uint32_t)__popcll(__ballot(f != 0));
    if (lane == b) cnt = c;
  }
  uint32_t excl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(excl, d, 64); if (lane >= (uint32_t)d) excl += t; }
  excl -= cnt;
  const uint32_t chunk_off = chunk_info[2 * blk_];
  uint32_t carry = chunk_info[2 * blk_ + 1];   // value of the next valid entry after the batch being processed
  // pass 2: batches from last to first so that the successor's value is known
  for (uint32_t bb = nb; bb-- > 0;) {
    const uint32_t f = staged[bb * 64u + lane];
    const unsigned long long valid = __ballot(f != 0);
    const unsigned long long ones = __ballot(f == 2);
    if (valid == 0ull) continue;
    const unsigned long long above = (lane == 63u) ? 0ull : (valid & ~((2ull << lane) - 1ull));
    uint32_t nxt = carry;
    if (above) nxt = (uint32_t)((ones >> (__ffsll((long long)above) - 1)) & 1ull);
    if (f != 0) {
      const uint32_t mine = (f == 2);
      const uint32_t at = chunk_off + rl(excl, bb) + lanes_below(valid);
      if (a.bits_out) a.bits_out[at] = (uint8_t)(mine == nxt); else rec[at] = pick(mine == nxt, e0, e1);
    }
    carry = (uint32_t)((ones >> (__ffsll((long long)valid) - 1)) & 1ull);
  }
}

// ---- the table stage on the device (see TableAtt) -------------------------------------------------------------
constexpr uint32_t kTablesLdsBins = 8192;   // alphabets up to this size have their normalised frequencies staged in LDS for the serialisation
struct TablesLds {
  uint64_t red[kTablesThreads / 64];
  uint32_t stage[4 * 1024];   // orientation summaries, 1024 blocks at a time
  uint32_t nextv;
  uint32_t fq[kTablesLdsBins + 2];   // normalised frequencies: a zero's token scans up to 64 entries ahead — from LDS, not 64 dependent global loads
};
__device__ __forceinline__ uint64_t block_sum(uint64_t v, uint64_t* red) {   // every thread receives the total
#pragma unroll
  for (int o = 32; o; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  uint64_t t = 0;
#pragma unroll
  for (uint32_t w = 0; w < kTablesThreads / 64; ++w) t += red[w];
  return t;
}
__device__ __forceinline__ uint64_t block_max(uint64_t v, uint64_t* red) {
#pragma unroll
  for (int o = 32; o; o >>= 1) { const uint64_t t = (uint64_t)__shfl_xor((unsigned long long)v, o, 64); v = t > v ? t : v; }
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  uint64_t t = 0;
#pragma unroll
  for (uint32_t w = 0; w < kTablesThreads / 64; ++w) t = red[w] > t ? red[w] : t;
  return t;
}
// the f32 "(c0 / len) * 256 + 0.5 → u16 → clamp(1, 255)" idiom of the metadata coders (mesh_normal_prediction.rs:147-150)
__device__ __forceinline__ uint32_t zero_probability_dev(uint64_t count_zero, float denominator) {
  const float p = ((float)count_zero / denominator) * 256.0f + 0.5f;
  const uint32_t q = (p != p || p <= 0.0f) ? 0u : (p >= 65535.0f ? 65535u : (uint32_t)p);   // Rust `as u16`
  return min(255u, max(1u, q));
}
// bytes symbol s contributes to the serialised table (rans.rs:199-228) and their values
__device__ __forceinline__ uint32_t table_token(const uint32_t* freq, uint32_t s, uint32_t& bytes) {
  const uint32_t f = freq[s];
  if (f) {
    const uint32_t extra = f >= (1u << 14) ? 2u : (f >= (1u << 6) ? 1u : 0u);
    bytes = ((f << 2) | extra) & 0xFFu;
    for (uint32_t b = 0; b < extra; ++b) bytes |= ((f >> (8u * (b + 1u) - 2u)) & 0xFFu) << (8u * (b + 1u));
    return 1u + extra;
  }
  // a zero: r = zeros that follow directly (capped at 64; the last symbol of the table is non-zero).  The serial loop emits
  // `(r << 2) | 3` and skips the r zeros when r ≤ 63, and `(64 << 2) | 3` truncated to a byte (= 3) WITHOUT skipping when
  // more follow (Q20) — so a zero emits a token iff it starts its run or at least 63 zeros follow it.
  uint32_t r = 0;
  while (r < 64u && freq[s + r + 1u] == 0u) ++r;
  const bool starts = s == 0u || freq[s - 1u] != 0u;
  if (!starts && r < 63u) return 0u;
  bytes = r >= 64u ? 3u : ((r << 2) | 3u);
  return 1u;
}
__device__ __forceinline__ void write_desc(ChainDesc* at, uint32_t kind, uint32_t precision, uint64_t n, const void* sym, uint32_t one_byte, const RansEntry* table,
                                           const uint32_t* batch_flags, uint32_t state0, uint8_t* out, uint64_t cap, uint32_t* out_len, uint32_t* ticks) {
  ChainDesc d;
  d.kind = kind; d.precision = precision; d.n = n; d.sym = sym; d.one_byte = one_byte; d.pad0 = 0; d.table = table; d.batch_flags = batch_flags;
  d.force_generic = 0; d.state0 = state0; d.out = out; d.cap = cap; d.out_len = out_len; d.ticks = ticks;
  *at = d;
}

__device__ void k_tables_body(const TableAtt& a, const uint32_t, const uint32_t) {
  __shared__ TablesLds lds;
  const uint32_t tid = threadIdx.x, T = kTablesThreads;
  const uint32_t* __restrict__ hist = a.hist;
  uint32_t* freq = a.freq;
  const uint32_t bins = a.bins;
  uint32_t err = 0;
  // ---- alphabet, bit_length, precision (symbol_coding.rs:46,118-141; Q11) ----
  uint64_t tot = 0, last = 0;
  for (uint32_t s = tid; s < bins; s += T) { const uint32_t h = hist[s]; tot += h; if (h) last = s + 1u; }
  const uint64_t total = block_sum(tot, lds.red);
  const uint32_t num_symbols = (uint32_t)block_max(last, lds.red);
  uint32_t precision = 12, bit_length = 1;
  if (num_symbols == 0) err = 1;
  if (!err) {
    const uint64_t nonzero = total - hist[0];
    const uint32_t bl = (nonzero ? 64u - (uint32_t)__builtin_clzll(nonzero) : 0u) + 1u;
    bit_length = min(18u, max(1u, bl));
    precision = bit_length <= 8 ? 12u : bit_length == 9 ? 13u : bit_length == 10 ? 15u : bit_length == 11 ? 16u : bit_length == 12 ? 18u : bit_length == 13 ? 19u : 20u;
  }
  const uint64_t target = 1ull << precision;
  // ---- normalisation (rans.rs:146-190) ----
  if (!err) {
    uint64_t part = 0;
    const double totd = (double)total;
    for (uint32_t s = tid; s < num_symbols; s += T) {
      const uint32_t h = hist[s];
      uint64_t nf = (uint64_t)(((double)h / totd) * (double)target + 0.5);
      if (nf == 0 && h) nf = 1;
      freq[s] = (uint32_t)nf;
      part += nf;
    }
    const uint64_t sum = block_sum(part, lds.red);   // (the barriers inside also publish freq[] to the block)
    if (sum < target) {
      // the deficit goes to the largest frequency, the LAST of equals (stable sort by key, Q12)
      uint64_t best = 0;
      for (uint32_t s = tid; s < num_symbols; s += T) { const uint64_t key = ((uint64_t)freq[s] << 32) | s; best = key > best ? key : best; }
      best = block_max(best, lds.red);
      if (tid == 0) freq[(uint32_t)best] += (uint32_t)(target - sum);
    } else if (sum > target) {
      // one off each of the `excess` largest, largest first, higher index first among equals: with T = the excess-th largest
      // value, everything above T loses one, and of the entries equal to T the ones with the highest indices
      const uint64_t excess = sum - target;
      if (excess > num_symbols) err = 2;
      if (!err) {
        // the two searches below sweep the frequencies ≈ 30 times: a thread keeps its (strided) entries in registers when the alphabet allows
        // (≤ 8 per thread = 8192 symbols: every default-width attribute), so a sweep costs a block reduction and no memory round trip
        constexpr uint32_t kKeep = 8;
        const bool kept = num_symbols <= kKeep * T;
        uint32_t fr[kKeep];
#pragma unroll
        for (uint32_t j = 0; j < kKeep; ++j) { const uint32_t s = tid + j * T; fr[j] = (kept && s < num_symbols) ? freq[s] : 0u; }
        // Both quantities are order statistics: thr = the excess-th largest frequency, ilo = the need-th largest INDEX among the entries equal to
        // thr.  Each is selected by its digits, 8 bits at a time from the top (three sweeps for keys below 2^24): a sweep counts the candidates'
        // current digit into 256 LDS bins, one thread walks the bins from the top to the one that holds the k-th, the candidates narrow to it.
        // (Rounds 2–3a bisected both: ≈ 30 block reductions, half of this kernel's 62 µs.)
        uint32_t* sel_hist = lds.stage;          // 256 bins (the orientation summaries use the array later)
        uint32_t* sel_out = lds.stage + 256;     // [0] the digit chosen, [1] how many of the k are still to be found inside it
        auto select_kth_largest = [&](auto&& key_of, uint32_t k, uint32_t& remaining_out) -> uint32_t {   // key_of(f, s, ok): the entry's key, ok = it takes part
          uint32_t prefix_bits = 0, remaining = k;
          for (int shift = 16; shift >= 0; shift -= 8) {
            for (uint32_t b = tid; b < 256u; b += T) sel_hist[b] = 0u;
            __syncthreads();
            const uint32_t hi_mask = shift == 16 ? 0u : (0xFFFFFFFFu << (shift + 8));   // the digits already fixed
            // (frequencies pile up on a few digits: the lanes of a wavefront that share a digit send ONE LDS atomic — a 1024-thread block adding
            //  to one bin entry by entry is ≈ 14 µs a sweep)
            auto visit = [&](uint32_t f, uint32_t s2, bool in_range) {
              bool ok = in_range;
              const uint32_t key = key_of(f, s2, ok);
              bool todo = in_range && ok && (key & hi_mask) == (prefix_bits & hi_mask);
              const uint32_t digit = (key >> shift) & 255u;
              for (unsigned long long m = __ballot(todo); m; m = __ballot(todo)) {
                const uint32_t d0 = (uint32_t)__shfl((int)digit, __ffsll((long long)m) - 1, 64);
                const unsigned long long same = __ballot(todo && digit == d0);
                if ((tid & 63u) == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(&sel_hist[d0], (uint32_t)__popcll(same));
                if (digit == d0) todo = false;
              }
            };
            if (kept) {
#pragma unroll
              for (uint32_t j = 0; j < kKeep; ++j) { const uint32_t s2 = tid + j * T; visit(fr[j], s2, s2 < num_symbols); }
            } else {
              for (uint32_t base = 0; base < num_symbols; base += T) { const uint32_t s2 = base + tid; visit(s2 < num_symbols ? freq[s2] : 0u, s2, s2 < num_symbols); }
            }
            __syncthreads();
            if (tid < 64u) {   // the bin that holds the k-th from the top: lane l owns bins 4l … 4l + 3, a suffix sum over the lanes finds the group
              const uint32_t c0 = sel_hist[4u * tid], c1 = sel_hist[4u * tid + 1u], c2 = sel_hist[4u * tid + 2u], c3 = sel_hist[4u * tid + 3u];
              const uint32_t g = c0 + c1 + c2 + c3;
              uint32_t suf = g;
#pragma unroll
              for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_down((int)suf, d, 64); if (tid + (uint32_t)d < 64u) suf += t; }
              const uint32_t above = suf - g, left = remaining;
              if (tid == 0 && suf < left) { sel_out[0] = 0u; sel_out[1] = left - above; }   // (fewer candidates than k: cannot happen for the two uses below)
              if (above < left && left <= suf) {
                uint32_t rem = left - above, d = 4u * tid + 3u;
                if (c3 < rem) { rem -= c3; --d; if (c2 < rem) { rem -= c2; --d; if (c1 < rem) { rem -= c1; --d; } } }
                sel_out[0] = d; sel_out[1] = rem;
              }
            }
            __syncthreads();
            prefix_bits |= sel_out[0] << shift;
            remaining = sel_out[1];
            __syncthreads();
          }
          remaining_out = remaining;
          return prefix_bits;
        };
        // (frequencies are at most 2^precision ≤ 2^20 and indices below 2^21: 24-bit keys)
        uint32_t need32 = 0;
        const uint32_t thr = select_kth_largest([&](uint32_t f, uint32_t, bool&) { return f; }, (uint32_t)excess, need32);
        if (thr == 0) err = 3;   // a zero would be decremented
        if (!err) {
          // need32 = excess − #(freq > thr): that many entries equal to thr lose one, from the highest index down
          uint32_t unused = 0;
          const uint32_t ilo = select_kth_largest([&](uint32_t f, uint32_t s2, bool& ok) { ok = f == thr; return s2; }, need32, unused);
          __syncthreads();
          for (uint32_t s = tid; s < num_symbols; s += T) { const uint32_t f = freq[s]; if (f > thr || (f == thr && s >= ilo)) freq[s] = f - 1u; }
        }
      }
    }
    __syncthreads();
    // the over-correction can drive an occurring symbol to 0: the reference coder does not terminate on such a table
    uint64_t bad = 0;
    for (uint32_t s = tid; s < num_symbols; s += T) bad += (hist[s] != 0u && freq[s] == 0u);
    if (block_sum(bad, lds.red) && !err) err = 4;
  }
  // ---- cumulative frequencies → coding records; the serialised table (rans.rs:192-228) ----
  uint32_t prefix = 0, hdr_len = 0;
  uint64_t rare = 0;
  uint8_t* hdr = a.hdr;
  if (!err) {
    uint8_t pre[8];
    pre[0] = 1; pre[1] = (uint8_t)bit_length;   // SymbolEncodingMethod::DirectCoded, bit_length
    prefix = 2;
    for (uint32_t v = num_symbols;;) { const uint8_t b = v & 0x7Fu; v >>= 7; if (v) pre[prefix++] = b | 0x80u; else { pre[prefix++] = b; break; } }
    if ((uint64_t)prefix + 3ull * num_symbols > a.hdr_cap) err = 5;
    if (!err && tid == 0) for (uint32_t k = 0; k < prefix; ++k) hdr[k] = pre[k];
  }
  if (!err) {
    uint64_t carry = 0;   // (header bytes << 32) | cumulative frequency
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const bool staged = num_symbols <= kTablesLdsBins;
    if (staged) {
      for (uint32_t s = tid; s < num_symbols; s += T) lds.fq[s] = freq[s];
      if (tid == 0) { lds.fq[num_symbols] = 1u; lds.fq[num_symbols + 1u] = 1u; }   // (never read past: the last symbol of a table is non-zero; a stop all the same)
      __syncthreads();
    }
    const uint32_t* fsrc = staged ? lds.fq : freq;
    for (uint32_t base = 0; base < num_symbols; base += T) {
      const uint32_t s = base + tid;
      uint32_t f = 0, nbytes = 0, bytes = 0;
      if (s < num_symbols) { f = fsrc[s]; nbytes = table_token(fsrc, s, bytes); }
      const uint64_t v = ((uint64_t)nbytes << 32) | f;
      uint64_t incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint64_t t = (uint64_t)__shfl_up((unsigned long long)incl, d, 64); if (lane >= (uint32_t)d) incl += t; }
      __syncthreads();
      if (lane == 63u) lds.red[wave] = incl;
      __syncthreads();
      uint64_t before = carry, chunk = 0;
#pragma unroll
      for (uint32_t w = 0; w < kTablesThreads / 64; ++w) { if (w < wave) before += lds.red[w]; chunk += lds.red[w]; }
      const uint64_t excl = before + incl - v;
      if (s < num_symbols) {
        a.rtable[s] = make_rans_entry(f, (uint32_t)excl, precision);
        uint8_t* at = hdr + prefix + (uint32_t)(excl >> 32);
        for (uint32_t k = 0; k < nbytes; ++k) at[k] = (uint8_t)(bytes >> (8u * k));
        if (f && ((uint64_t)f << 8) < target) rare += hist[s];
      }
      carry += chunk;
    }
    hdr_len = prefix + (uint32_t)(carry >> 32);
    rare = block_sum(rare, lds.red);
  }
  // ---- the metadata stream of the prediction scheme ----
  uint32_t zero_prob = 0, aux_count = 0;
  if (a.aux_kind == 1) {           // normal flips: mesh_normal_prediction.rs:147-150 (count of `false` in small[2])
    aux_count = a.n_entries;
    zero_prob = zero_probability_dev(a.small[2], (float)a.n_entries);
  } else if (a.aux_kind == 2) {    // orientations: stitch the per-block summaries {count, first, last, transitions} (…texture_coordinates.rs:224-253, Q10)
    // Every thread owns a run of consecutive chunks; what crosses the runs — the compact offset, the value of the last valid flag before
    // the run, the value of the first valid flag after it — comes from block-wide scans (the serial loops this replaces walked all chunks
    // on one lane: ≈ 0.2 ms for the 1221 chunks of the 10M-triangle workload, most of the table stage).
    const uint32_t nb = a.summary_blocks;
    const uint32_t per = (nb + T - 1u) / T;
    const uint32_t b0 = min(nb, tid * per), b1 = min(nb, b0 + per);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint64_t cnt = 0, tr = 0;
    uint32_t has = 0, firstv = 1, lastv = 1;
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t c = a.summary[4u * b];
      if (!c) continue;
      const uint32_t f = a.summary[4u * b + 1u], l = a.summary[4u * b + 2u];
      if (has) tr += f != lastv; else firstv = f;
      tr += a.summary[4u * b + 3u];
      lastv = l; has = 1; cnt += c;
    }
    // exclusive scans over the threads: counts (sum), "last valid value so far" (forward), "first valid value from here on" (backward)
    uint64_t inc_cnt = cnt;
    uint32_t fwd = has ? (2u | lastv) : 0u;     // bit 1: some chunk up to and including this thread holds a flag; bit 0: the last such value
    uint32_t bwd = has ? (2u | firstv) : 0u;    // the same from the other end
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t c2 = (uint64_t)__shfl_up((unsigned long long)inc_cnt, d, 64);
      const uint32_t f2 = (uint32_t)__shfl_up((int)fwd, d, 64), g2 = (uint32_t)__shfl_down((int)bwd, d, 64);
      if (lane >= (uint32_t)d) { inc_cnt += c2; if (!(fwd & 2u)) fwd = f2; }
      if (lane + (uint32_t)d < 64u) { if (!(bwd & 2u)) bwd = g2; }
    }
    __syncthreads();
    if (lane == 63u) { lds.red[wave] = inc_cnt; lds.stage[wave] = fwd; }
    if (lane == 0u) lds.stage[64u + wave] = bwd;
    __syncthreads();
    uint64_t before = 0, total_cnt = 0;
    uint32_t prev = 0, next = 0;   // (2 | value) of the last valid flag before this thread's wave / the first one after it
#pragma unroll
    for (uint32_t w = 0; w < kTablesThreads / 64; ++w) {
      total_cnt += lds.red[w];
      if (w < wave) { before += lds.red[w]; if (lds.stage[w] & 2u) prev = lds.stage[w]; }
    }
    for (uint32_t w = kTablesThreads / 64; w-- > wave + 1u;) if (lds.stage[64u + w] & 2u) next = lds.stage[64u + w];
    // within the wave: the neighbours' inclusive results are the exclusive ones
    const uint32_t f_ex = (uint32_t)__shfl_up((int)fwd, 1, 64), g_ex = (uint32_t)__shfl_down((int)bwd, 1, 64);
    const uint32_t prev_t = (lane > 0u && (f_ex & 2u)) ? f_ex : prev;
    const uint32_t next_t = (lane < 63u && (g_ex & 2u)) ? g_ex : next;
    const uint32_t last_before = (prev_t & 2u) ? (prev_t & 1u) : 1u;   // `last` starts as true (…texture_coordinates.rs:224-235)
    const uint32_t first_after = (next_t & 2u) ? (next_t & 1u) : 1u;   // `true` after the last valid entry
    if (has && firstv != last_before) ++tr;
    uint32_t off = (uint32_t)(before + inc_cnt - cnt);
    for (uint32_t b = b0; b < b1; ++b) { a.chunk_info[2u * b] = off; off += a.summary[4u * b]; }
    uint32_t nextv = first_after;
    for (uint32_t b = b1; b-- > b0;) { a.chunk_info[2u * b + 1u] = nextv; if (a.summary[4u * b]) nextv = a.summary[4u * b + 1u]; }
    const uint64_t trans = block_sum(tr, lds.red);
    const uint64_t len = total_cnt;
    aux_count = (uint32_t)len;
    zero_prob = zero_probability_dev(trans, (float)len + 0.001f);
  }
  if (tid == 0) {
    a.small[6] = hdr_len;
    a.small[7] = err;
    a.small[12] = precision;   // (read by the host-core chains; a device chain overwrites it with its clock)
    a.small[14] = zero_prob;
    a.small[15] = aux_count;
    // which step the stream's walker uses: the one-byte step pays off when few batches of 64 hold a rare symbol (f < 2^(P-8))
    const double clean = pow(1.0 - (double)rare / (double)max((uint64_t)1, a.n_sym), 64.0);
    write_desc(a.desc, 0u, precision, err ? 0ull : a.n_sym, a.sym, clean > 0.8 ? 1u : 0u, a.rec, a.batch_flags, 4u << precision, a.out, a.out_cap, a.small + 8, a.small + 12);
    if (a.aux_kind) {
      // rABS (rans.rs:91-108): bit 1 codes with f1 = 256 - p0 and offset 0, bit 0 with p0 and offset f1
      const uint32_t p0 = zero_prob, f1 = 256u - p0;
      a.aux_entries[0] = make_rans_entry(p0, f1, 8);
      a.aux_entries[1] = make_rans_entry(f1, 0, 8);
      write_desc(a.aux_desc, a.aux_kind, 8u, aux_count, nullptr, 0u, a.aux_rec, a.aux_flags, 4096u, a.aux_out, a.aux_cap, a.small + 10, a.small + 13);
    }
    if (a.hdr_desc) write_desc(a.hdr_desc, 3u, 0u, 0ull, nullptr, 0u, nullptr, nullptr, 0u, a.hdr, a.hdr_cap, a.small + 6, nullptr);
  }
}

// ---- the chain -------------------------------------------------------------------------------------
// One step of the recurrence on the scalar unit (x, the record's m b d c t in SGPRs; s100 s101 temporaries):
//   general step, any renormalisation (10 scalar + 1 writelane):
//     q0 = mulhi(x, m) >> b ; sh = (BIAS - clz(q0)) & 24 ; park x in lane j ; x = (x >> sh) + (q0 >> sh)·d + c
//   one-byte step, for batches in which no symbol can renormalise by more than one byte (9 scalar + 1 writelane):
//     sh = (x ≥ t) ? 8 : 0 ; park x ; x >>= sh ; x = x + (mulhi(x, m) >> b)·d + c        (divide AFTER renormalising)
// The steady state is generated assembly (scripts/gen_walker_asm.py → dmi_walker_asm.inc); see chain_walker below.
// A stream is owned by a PAIR of wavefronts that talk through LDS (four pairs per workgroup, see k_chains):
//   the walker   runs the recurrence on its scalar unit and parks pre-renormalisation states of a batch in a VGPR (every
//                state, lane j = step j; or every 4th in a sparse launch), then drops them into a ring slot and bumps `produced`;
//   the emitter  picks the slot up, derives every step's byte count from (state, frequency), takes wavefront
//                    prefix sums with ballots + mbcnt and stores the bytes; it also touches the records a few
//                    batches AHEAD of the walker so that the walker's scalar loads hit L2.
// LDS operations of one wavefront execute in order, so "slot, then counter" needs no fence; the walker only looks at
// `consumed` when its cached copy says the ring could be full (once per kRing batches).
// The record buffer is padded with kChainPad records past n, so chunk and batch prefetches may run ahead freely.
constexpr uint32_t kRing = 8;
constexpr uint32_t kAhead = 4;   // batches between the emitter's position and the records it pulls into L2
// One walker/emitter pair's LDS.  `produced` / `consumed` count ring slots over the pair's whole life (all the streams it
// serves): a stream of nb batches takes nb + 1 slots, the last one carrying {lane 0: final state, lane 1: the next stream
// of the pair} — hand-over between streams needs no other synchronisation.
struct ChainShared {
  uint32_t ring[kRing][64];
  uint32_t produced, consumed, pad[2];
};
constexpr uint32_t kNoStream = 0xFFFFFFFFu;
typedef const RansEntry __attribute__((address_space(1))) * grec_t;   // global (not flat): flat accesses count in lgkmcnt too
typedef volatile ChainShared __attribute__((address_space(3))) * lds_shared_t;   // LDS address space ⇒ ds_read/ds_write

#include "dmi_walker_asm.inc"
// Two parking modes.  Dense: the state before every step is parked (lane j = step j) and the emitter only sheds bytes.  Sparse: the
// state before every kPark-th step only (one v_writelane per kPark steps: the walker is bound by its instruction count — 266 → 251 ms
// on the 15M-symbol chain); lane l of a ring slot holds the state before step l·kPark and the emitter re-runs the steps in between
// with the same exact arithmetic.  The heavier emitter costs more than it saves once pairs share SIMDs (every walker then has another
// pair's emitter on its SIMD: 1024-mesh batch 16.1 → 19.3 ms), so a launch is sparse only when its streams fit one per CU.
constexpr uint32_t kPark = DMI_WALKER_PARK;
constexpr uint32_t kParkLanes = 64u / kPark;

// The walker.  Runs of consecutive full batches without a flagged symbol execute in one hand-scheduled assembly loop
// (scripts/gen_walker_asm.py → dmi_walker_asm.inc: records double-buffered in two fixed 40-SGPR sets, the next chunk
// requested right after each wait, ≈ 20 instructions of hand-off per 64 steps); a flagged batch (frequency-1 symbol; for
// ONE_BYTE streams also a rare symbol) and the tail batch take the generic divide loop here.
// K = the pair's slot counter (see ChainShared); returns the final state.
template <uint32_t BIAS, bool ONE_BYTE, uint32_t PARK>
__device__ uint32_t chain_walker(const ChainDesc& d, uint32_t lane, lds_shared_t sh, uint32_t& K, uint32_t& consumed_seen) {
  constexpr uint32_t bias = BIAS;                 // 29 (rANS, threshold f·2^10) or 27 (rABS, f·2^12)
  const uint64_t n = d.n;
  const RansEntry* __restrict__ rec = d.table;
  uint32_t x = d.state0;
  typedef const uint32_t __attribute__((address_space(4))) * const_u32_t;
  const_u32_t flags = (const_u32_t)(uintptr_t)d.batch_flags;   // nullable
  const grec_t grec = (grec_t)(uintptr_t)rec;
  const uint32_t nb = (uint32_t)((n + 63) >> 6), full = (uint32_t)(n >> 6);
  constexpr uint32_t flag_mask = ONE_BYTE ? 3u : 1u;   // bit 0: frequency-1 symbol, bit 1: rare symbol (may renormalise > 1 byte)
  const uint32_t ring_lane = (uint32_t)(uintptr_t)&sh->ring[0][lane];
  const uint32_t produced_at = (uint32_t)(uintptr_t)&sh->produced, consumed_at = (uint32_t)(uintptr_t)&sh->consumed;
  uint32_t k = 0;                                 // batch of this stream
  while (k < nb) {
    const uint32_t flag = d.force_generic ? 1u : (flags ? flags[k] : 1u);
    if (k < full && !(flag & flag_mask)) {
      uint32_t left = full - k, vtmp, parked;
      const uint64_t rec_at = (uint64_t)(uintptr_t)(rec + (uint64_t)k * 64u), flag_at = (uint64_t)(uintptr_t)(d.batch_flags + k);
#define DMI_WALKER_RUN(BODY)                                                                                                \
  asm volatile(BODY : "+s"(x), "+s"(K), "+s"(left), "+s"(consumed_seen), "=&v"(vtmp), "=&v"(parked)                          \
               : "s"(rec_at), "s"(flag_at), "v"(ring_lane), "v"(produced_at), "v"(consumed_at) : DMI_WALKER_ASM_CLOBBERS)
      if (ONE_BYTE) { if (PARK == 1) DMI_WALKER_RUN(DMI_WALKER_ASM_ONE_BYTE_P1); else DMI_WALKER_RUN(DMI_WALKER_ASM_ONE_BYTE_P4); }
      else { if (PARK == 1) DMI_WALKER_RUN(DMI_WALKER_ASM_GENERAL_RANS_P1); else DMI_WALKER_RUN(DMI_WALKER_ASM_GENERAL_RANS_P4); }
#undef DMI_WALKER_RUN
      k = full - left;
      continue;
    }
    if (ONE_BYTE && BIAS == 29u && k < full && (flag & 3u) == 2u) {
      // a batch of a one-byte stream that holds a rare symbol (but no frequency-1 one): ONE batch through the general assembly step
      // (≈ 2.9k clocks) instead of the divide loop below (≈ 7k)
      uint32_t left = 1u, vtmp, parked;
      const uint64_t rec_at = (uint64_t)(uintptr_t)(rec + (uint64_t)k * 64u), flag_at = (uint64_t)(uintptr_t)(d.batch_flags + k);
#define DMI_WALKER_RUN(BODY)                                                                                                \
  asm volatile(BODY : "+s"(x), "+s"(K), "+s"(left), "+s"(consumed_seen), "=&v"(vtmp), "=&v"(parked)                          \
               : "s"(rec_at), "s"(flag_at), "v"(ring_lane), "v"(produced_at), "v"(consumed_at) : DMI_WALKER_ASM_CLOBBERS)
      if (PARK == 1) DMI_WALKER_RUN(DMI_WALKER_ASM_GENERAL_RANS_P1); else DMI_WALKER_RUN(DMI_WALKER_ASM_GENERAL_RANS_P4);
#undef DMI_WALKER_RUN
      k += 1u - left;
      continue;
    }
    // generic batch (a frequency-1 symbol, or the partial last batch): one record per lane through the vector path, broadcast step
    // by step; the quotient by multiply-high like the assembly step, or the state itself for a frequency-1 record
    const uint64_t base = (uint64_t)k << 6;
    const uint32_t cnt = (uint32_t)min((uint64_t)64, n - base);
    uint32_t parked = 0;
    x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    const uint32_t mm = grec[base + lane].m, mb = grec[base + lane].b, md = grec[base + lane].d, mc = grec[base + lane].c;
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t mj = rl(mm, j), bj = rl(mb, j), dj = rl(md, j), cj = rl(mc, j);
      const uint32_t q0 = (bj & 0x100u) ? x : (__umulhi(x, mj) >> (bj & 31u));
      const uint32_t shf = (bias - (uint32_t)__builtin_clz(q0)) & 0x18u;
      if ((j % PARK) == 0u && lane == j / PARK) parked = x;   // parking mode of the launch, like the assembly loop
      x = (x >> shf) + (q0 >> shf) * dj + cj;
    }
    while (K - consumed_seen >= kRing) {          // ring full (rare: the emitter is ≈6× faster than the walker)
      consumed_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->consumed);
      if (K - consumed_seen >= kRing) __builtin_amdgcn_s_sleep(4);
    }
    sh->ring[K & (kRing - 1u)][lane] = parked;
    sh->produced = K + 1u;
    ++K;
    ++k;
  }
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
}

// The emitter of a pair, dense parking (one parked state per step); returns the pair's next stream (from the stream's closing slot).
template <uint32_t BIAS>
__device__ uint32_t chain_emitter_dense(const ChainDesc& d, uint32_t lane, lds_shared_t sh, uint32_t& K) {
  constexpr uint32_t thr_shift = 39u - BIAS;      // 10 / 12
  const uint32_t P = d.precision;
  const uint64_t n = d.n;
  const grec_t grec = (grec_t)(uintptr_t)d.table;
  typedef uint8_t __attribute__((address_space(1))) * gbyte_t;
  uint64_t pos = 0;
  uint32_t err = 0, touched = 0;
  uint32_t me_d = grec[lane].d;                    // lane j's record of the batch being emitted (only d is needed)
  uint32_t far = 0;
#pragma unroll
  for (uint32_t a = 1; a < kAhead; ++a) touched ^= grec[(uint64_t)a * 64u + lane].m;
  for (uint64_t base = 0; base < n; base += 64, ++K) {
    const uint32_t cnt = (uint32_t)min((uint64_t)64, n - base);
    const uint32_t my_d = me_d;                    // requested one batch ago
    touched ^= far;
    me_d = grec[base + 64u + lane].d;
    far = grec[base + (uint64_t)kAhead * 64u + lane].m;   // one dword per record = every line of that batch
    while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)sh->produced) - K) <= 0) __builtin_amdgcn_s_sleep(2);
    const uint32_t parked = sh->ring[K & (kRing - 1u)][lane];
    sh->consumed = K + 1u;                         // (LDS is in order: the slot read above is performed first)
    // bytes of the 64 steps: lane j re-derives its byte count from (parked state, frequency)
    // (f ≤ 2^20 ⇒ f << thr_shift < 2^32; parked < 2^30; at most 3 bytes per step)
    const uint32_t thr = (lane < cnt) ? (((1u << P) - my_d) << thr_shift) : 0xFFFFFFFFu;
    const bool b0 = parked >= thr, b1 = (parked >> 8) >= thr, b2 = (parked >> 16) >= thr;
    const unsigned long long m0 = __ballot(b0), m1 = __ballot(b1), m2 = __ballot(b2);
    const uint32_t before = lanes_below(m0) + lanes_below(m1) + lanes_below(m2);
    const uint32_t total = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
    if (pos + total > d.cap) err = 2;
    if (!err) {
      gbyte_t at = (gbyte_t)(uintptr_t)d.out + pos + before;
      if (b0) at[0] = (uint8_t)parked;
      if (b1) at[1] = (uint8_t)(parked >> 8);
      if (b2) at[2] = (uint8_t)(parked >> 16);
      pos += total;
    }
  }
  // the stream's closing slot
  while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)sh->produced) - K) <= 0) __builtin_amdgcn_s_sleep(2);
  const uint32_t closing = sh->ring[K & (kRing - 1u)][lane];
  sh->consumed = K + 1u;
  ++K;
  const uint32_t x = rl(closing, 0), next = rl(closing, 1);
  if (lane == 0) {
    if (!err) pos += flush_state(x - d.state0, d.out, pos, d.cap, err);
    if ((touched ^ far) == 0x9E3779B9u && pos == ~0ull) err = 3;   // keeps the look-ahead loads alive; never true
    d.out_len[0] = (uint32_t)pos;
    d.out_len[1] = err;
  }
  return next;
}

// The emitter of a pair, sparse parking; returns the pair's next stream (from the stream's closing slot).
// Lane l < kParkLanes owns steps l·kPark … l·kPark + kPark - 1 of a batch: from the parked state it replays them (renormalisation
// byte count from the thresholds f·2^T, f·2^(T+8), f·2^(T+16); exact quotient by multiply-high, or the state itself for a
// frequency-1 record; x' = xs + q·d + c — the walker's arithmetic), keeping every step's bytes; a prefix sum over the lanes
// gives each lane its output offset.  The records of the NEXT batch are requested before this one is processed.
struct EmitRec { uint32_t m, b, d, c; };
template <uint32_t BIAS>
__device__ uint32_t chain_emitter_sparse(const ChainDesc& d, uint32_t lane, lds_shared_t sh, uint32_t& K) {
  constexpr uint32_t thr_shift = 39u - BIAS;      // 10 / 12
  const uint32_t P = d.precision;
  const uint64_t n = d.n;
  const grec_t grec = (grec_t)(uintptr_t)d.table;
  typedef uint8_t __attribute__((address_space(1))) * gbyte_t;
  uint64_t pos = 0;
  uint32_t err = 0, touched = 0;
  const bool owner = lane < kParkLanes;
  EmitRec cur[kPark], nxt[kPark];
#pragma unroll
  for (uint32_t k = 0; k < kPark; ++k) {
    const grec_t r = grec + ((owner ? lane : 0u) * kPark + k);   // (the record buffer is padded: reads past n are harmless)
    nxt[k] = EmitRec{r->m, r->b, r->d, r->c};
  }
  uint32_t far = 0;
#pragma unroll
  for (uint32_t a = 1; a < kAhead; ++a) touched ^= grec[(uint64_t)a * 64u + lane].m;
  for (uint64_t base = 0; base < n; base += 64, ++K) {
    const uint32_t cnt = (uint32_t)min((uint64_t)64, n - base);
    touched ^= far;
#pragma unroll
    for (uint32_t k = 0; k < kPark; ++k) cur[k] = nxt[k];   // requested one batch ago
#pragma unroll
    for (uint32_t k = 0; k < kPark; ++k) {
      const grec_t r = grec + (base + 64u + (owner ? lane : 0u) * kPark + k);
      nxt[k] = EmitRec{r->m, r->b, r->d, r->c};
    }
    far = grec[base + (uint64_t)kAhead * 64u + lane].m;   // one dword per record = every line of that batch
    while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)sh->produced) - K) <= 0) __builtin_amdgcn_s_sleep(2);
    uint32_t x = sh->ring[K & (kRing - 1u)][lane];
    sh->consumed = K + 1u;                         // (LDS is in order: the slot read above is performed first)
    // replay: bytes[k] = the (≤ 3) low bytes step k sheds, counts = their numbers, 2 bits each
    uint32_t bytes[kPark], counts = 0, mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPark; ++k) {
      const bool live = owner && lane * kPark + k < cnt;
      const uint32_t f = (1u << P) - cur[k].d;
      const uint32_t thr = f << thr_shift;   // (f ≤ 2^20 ⇒ < 2^32; x < 2^30; at most 3 bytes per step)
      const uint32_t nb = live ? (uint32_t)(x >= thr) + (uint32_t)((x >> 8) >= thr) + (uint32_t)((x >> 16) >= thr) : 0u;
      bytes[k] = x;
      counts |= nb << (2u * k);
      mine += nb;
      const uint32_t xs = x >> (8u * nb);
      const uint32_t q = (cur[k].b & 0x100u) ? xs : (__umulhi(xs, cur[k].m) >> (cur[k].b & 31u));
      x = xs + q * cur[k].d + cur[k].c;
    }
    uint32_t incl = mine;   // prefix over the owning lanes
#pragma unroll
    for (uint32_t dlt = 1; dlt < kParkLanes; dlt <<= 1) { const uint32_t t = __shfl_up(incl, dlt, 64); if (lane >= dlt) incl += t; }
    const uint32_t total = rl(incl, kParkLanes - 1u);
    if (pos + total > d.cap) err = 2;
    if (!err) {
      gbyte_t at = (gbyte_t)(uintptr_t)d.out + pos + (incl - mine);
#pragma unroll
      for (uint32_t k = 0; k < kPark; ++k) {
        const uint32_t nb = (counts >> (2u * k)) & 3u;
        if (nb > 0) at[0] = (uint8_t)bytes[k];
        if (nb > 1) at[1] = (uint8_t)(bytes[k] >> 8);
        if (nb > 2) at[2] = (uint8_t)(bytes[k] >> 16);
        at += nb;
      }
      pos += total;
    }
  }
  // the stream's closing slot
  while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)sh->produced) - K) <= 0) __builtin_amdgcn_s_sleep(2);
  const uint32_t closing = sh->ring[K & (kRing - 1u)][lane];
  sh->consumed = K + 1u;
  ++K;
  const uint32_t x = rl(closing, 0), next = rl(closing, 1);
  if (lane == 0) {
    if (!err) pos += flush_state(x - d.state0, d.out, pos, d.cap, err);
    if ((touched ^ far ^ nxt[0].m) == 0x9E3779B9u && pos == ~0ull) err = 3;   // keeps the look-ahead loads alive; never true
    d.out_len[0] = (uint32_t)pos;
    d.out_len[1] = err;
  }
  return next;
}

// Every rANS / rABS stream of a launch.  A workgroup is FOUR walker/emitter pairs: a wavefront issues one instruction per
// ≈ 4 clocks and the scalar unit of a SIMD serves one wavefront per clock turn, so two walkers on one SIMD run at half
// speed each — the walkers are wavefronts 0..3 of the workgroup, which the dispatcher places on the four SIMDs of one CU
// (wavefront w + 4 shares the SIMD of wavefront w: scripts/probes/simd_probe.hip), and pair p's emitter is wavefront
// 4 + (p + 1) % 4, i.e. on another SIMD than its walker.  With ≤ 256 workgroups (one per CU) every walker of the chip
// has a SIMD's scalar unit to itself.  Streams are served longest first (`order`): the first 4·gridDim.x are dealt
// statically (stream s → workgroup s mod G, pair s / G: the longest G streams sit alone on their CUs' first SIMD), the rest
// is pulled through the counter `next_stream` (zeroed by the launcher) when a pair finishes a stream — the launch lasts max(longest stream, total steps / walkers)
// instead of being at the mercy of the dispatcher's placement (1024-mesh batch: 13.5 → … ms).
constexpr uint32_t kChainPairs = 4;
// PAIRS = pairs of a workgroup that take streams: 4 (every SIMD hosts a walker and another pair's emitter), or 2 — pairs 0 and 2, whose
// walkers sit on the first and third SIMD and whose emitters on the second and fourth, so that no wavefront shares a SIMD (sparse launches).
template <uint32_t PARK, uint32_t PAIRS>
__global__ __launch_bounds__(512) void k_chains(const ChainDesc* __restrict__ descs, const uint32_t* __restrict__ order, uint32_t n_streams,
                                                uint32_t* __restrict__ next_stream) {
  __shared__ ChainShared shared[kChainPairs];
  if (threadIdx.x < kChainPairs) { shared[threadIdx.x].produced = 0; shared[threadIdx.x].consumed = 0; }
  __syncthreads();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t lane = threadIdx.x & 63u;
  const bool walker = wave < kChainPairs;
  const uint32_t pair = walker ? wave : ((wave - kChainPairs) + kChainPairs - 1u) % kChainPairs;   // emitter of pair p = wavefront 4 + (p + 1) % 4
  const lds_shared_t sh = (lds_shared_t)&shared[pair];
  if (PAIRS == 2u && (pair & 1u)) return;
  const uint32_t first = (PAIRS == 2u ? pair / 2u : pair) * gridDim.x + blockIdx.x;
  if (first >= n_streams) return;
  // (two loops with their own variables: the walker's live in SGPRs for the assembly, the emitter's do not)
  if (walker) {
    uint32_t K = 0, consumed_seen = 0, sid = first;
    while (sid != kNoStream) {
      const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(order ? order[sid] : sid));
      const ChainDesc d = descs[at];
      uint32_t x;
      if (d.kind == 0) x = d.one_byte ? chain_walker<29u, true, PARK>(d, lane, sh, K, consumed_seen) : chain_walker<29u, false, PARK>(d, lane, sh, K, consumed_seen);
      else x = chain_walker<27u, true, PARK>(d, lane, sh, K, consumed_seen);   // rABS renormalises with a single `if` (rans.rs:97): never more than one byte
      uint32_t next = 0;
      if (lane == 0) next = atomicAdd(next_stream, 1u);
      next = (uint32_t)__builtin_amdgcn_readfirstlane((int)next) + PAIRS * gridDim.x;
      if (next >= n_streams) next = kNoStream;
      while (K - consumed_seen >= kRing) {
        consumed_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->consumed);
        if (K - consumed_seen >= kRing) __builtin_amdgcn_s_sleep(4);
      }
      sh->ring[K & (kRing - 1u)][lane] = lane == 0 ? x : next;
      sh->produced = K + 1u;
      ++K;
      sid = next;
    }
  } else {
    uint32_t K = 0, sid = first;
    while (sid != kNoStream) {
      const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(order ? order[sid] : sid));
      const ChainDesc d = descs[at];
      const uint64_t t0 = wall_clock64();   // 100 MHz constant-rate counter
      if (PARK == 1) sid = d.kind == 0 ? chain_emitter_dense<29u>(d, lane, sh, K) : chain_emitter_dense<27u>(d, lane, sh, K);
      else sid = d.kind == 0 ? chain_emitter_sparse<29u>(d, lane, sh, K) : chain_emitter_sparse<27u>(d, lane, sh, K);
      if (lane == 0 && d.ticks) d.ticks[0] = (uint32_t)(wall_clock64() - t0);
    }
  }
}

// ---- batch read-back: every stream's bytes packed into one arena (one D2H for a whole batch of meshes) -----------
// table[k] = {offset (16-byte aligned), length, error flag} of stream k, table[n] = {total, 0, 0}.
__global__ __launch_bounds__(1024) void k_pack_offsets(const ChainDesc* __restrict__ descs, uint32_t n, PackEntry* __restrict__ table) {
  __shared__ uint64_t wave_sum[16];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t k = base + threadIdx.x;
    uint32_t len = 0, err = 0;
    if (k < n) { len = descs[k].out_len[0]; err = descs[k].out_len[1]; }
    const uint64_t padded = ((uint64_t)len + 15ull) & ~15ull;
    uint64_t incl = padded;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t t = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= (uint32_t)d) incl += t; }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t before = carry;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_sum[w];
    if (k < n) { table[k].offset = before + incl - padded; table[k].len = len; table[k].err = err; }
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) { table[n].offset = carry; table[n].len = 0; table[n].err = 0; }
}
constexpr uint32_t kPackSplit = 8;   // blocks per stream
__global__ __launch_bounds__(256) void k_pack_copy(const ChainDesc* __restrict__ descs, const PackEntry* __restrict__ table, uint8_t* __restrict__ arena) {
  const uint32_t k = blockIdx.x;
  const uint32_t len = table[k].len;
  const uint64_t vecs = ((uint64_t)len + 15ull) >> 4;   // whole 16-byte words (the source buffers are padded allocations)
  const uint64_t lo = vecs * blockIdx.y / kPackSplit, hi = vecs * (blockIdx.y + 1) / kPackSplit;
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(descs[k].out);
  uint4* __restrict__ dst = reinterpret_cast<uint4*>(arena + table[k].offset);
  for (uint64_t v = lo + threadIdx.x; v < hi; v += 256) dst[v] = src[v];
}

// Generic gather of device ranges into one arena (batch read-back of the jobs' slabs): 16-byte words, 256-aligned pieces.
__global__ __launch_bounds__(256) void k_copy_items(const CopyItem* __restrict__ items, uint8_t* __restrict__ arena) {
  const CopyItem it = items[blockIdx.x];
  const uint64_t vecs = it.bytes >> 4;
  const uint64_t lo = vecs * blockIdx.y / kPackSplit, hi = vecs * (blockIdx.y + 1) / kPackSplit;
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(it.src);
  uint4* __restrict__ dst = reinterpret_cast<uint4*>(arena + it.dst_offset);
  for (uint64_t v = lo + threadIdx.x; v < hi; v += 256) dst[v] = src[v];
}

// The inverse: pieces of one staging arena scattered to their device destinations (batch upload of the jobs' coding tables).
__global__ __launch_bounds__(256) void k_scatter_items(const CopyItem* __restrict__ items, const uint8_t* __restrict__ arena) {
  const CopyItem it = items[blockIdx.x];   // src = device destination, dst_offset = offset of the piece in the arena
  const uint64_t vecs = it.bytes >> 4;
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(arena + it.dst_offset);
  uint4* __restrict__ dst = reinterpret_cast<uint4*>(const_cast<void*>(it.src));
  for (uint64_t v = threadIdx.x; v < vecs; v += 256) dst[v] = src[v];
}

__global__ __launch_bounds__(256) void k_clear_items(const CopyItem* __restrict__ items) {
  const CopyItem it = items[blockIdx.x];
  uint32_t* __restrict__ d = reinterpret_cast<uint32_t*>(const_cast<void*>(it.src));
  const uint64_t n = it.bytes >> 2;
  for (uint64_t v = threadIdx.x; v < n; v += 256) d[v] = 0u;
}

// the same for a handful of ranges passed BY VALUE (job creation of one mesh: eleven hipMemsetAsync calls were eleven launches of ≈ 5 µs back to back)
__global__ __launch_bounds__(256) void k_clear_ranges(const ClearRanges r) {
  uint32_t* __restrict__ d = static_cast<uint32_t*>(r.p[blockIdx.y]);
  const uint64_t bytes = r.bytes[blockIdx.y], n = bytes >> 2;
  const uint32_t b = r.value[blockIdx.y], w = b * 0x01010101u;
  for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (uint64_t)gridDim.x * 256) d[v] = w;
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 3u)) reinterpret_cast<uint8_t*>(d)[(n << 2) + threadIdx.x] = (uint8_t)b;   // (the last bytes of a range that is no multiple of 4)
}

inline uint32_t grid256(uint64_t n) { uint64_t g = (n + 255) / 256; return (uint32_t)(g > 4096 ? 4096 : (g ? g : 1)); }

#define DMI_PREP_KERNEL(NAME, BODY, ARGS, THREADS)                                                                                    \
  __global__ __launch_bounds__(THREADS) void NAME(ARGS a) { BODY(a, blockIdx.x, gridDim.x); }                                         \
  __global__ __launch_bounds__(THREADS) void NAME##_multi(const ARGS* __restrict__ items, const uint2* __restrict__ block_info,       \
                                                          const uint32_t* __restrict__ item_blocks) {                                 \
    const uint2 bi = block_info[blockIdx.x];                                                                                          \
    BODY(items[bi.x], bi.y, item_blocks[bi.x]);                                                                                        \
  }
DMI_PREP_KERNEL(k_rans_prep, k_rans_prep_body, RansPrepArgs, 256)
DMI_PREP_KERNEL(k_batch_flags, k_batch_flags_body, BatchFlagsArgs, 256)
DMI_PREP_KERNEL(k_bits_prep, k_bits_prep_body, BitsPrepArgs, 256)
DMI_PREP_KERNEL(k_orient_prep, k_orient_prep_body, OrientPrepArgs, 64)
DMI_PREP_KERNEL(k_tables, k_tables_body, TableAtt, kTablesThreads)
// the table stage of every attribute of ONE job in one launch (block b = attribute b): three dependent 100 µs launches otherwise
__global__ __launch_bounds__(kTablesThreads) void k_tables_group(TableGroup g) { k_tables_body(g.a[blockIdx.x], 0u, 1u); }

template <class Args>
void emit_prep(int id, int level, const Args& a, uint32_t blocks, hipStream_t s) {
  static_assert(sizeof(Args) <= sizeof(KernelStep::args), "KernelStep::args too small");
  if (blocks == 0) return;
  KernelStep st{};
  st.id = id; st.level = level; st.blocks = blocks; st.lds = 0; st.args_size = (uint32_t)sizeof(Args);
  std::memcpy(st.args, &a, sizeof(Args));
  if (!step_sink_push(st)) launch_step(st, s);
}

}  // namespace

void launch_prep_step(const KernelStep& st, hipStream_t s) {
  switch (st.id) {
    case K_RANS_PREP: hipLaunchKernelGGL(k_rans_prep, st.blocks, 256, 0, s, *reinterpret_cast<const RansPrepArgs*>(st.args)); break;
    case K_BATCH_FLAGS: hipLaunchKernelGGL(k_batch_flags, st.blocks, 256, 0, s, *reinterpret_cast<const BatchFlagsArgs*>(st.args)); break;
    case K_BITS_PREP: hipLaunchKernelGGL(k_bits_prep, st.blocks, 256, 0, s, *reinterpret_cast<const BitsPrepArgs*>(st.args)); break;
    case K_ORIENT_PREP: hipLaunchKernelGGL(k_orient_prep, st.blocks, 64, 0, s, *reinterpret_cast<const OrientPrepArgs*>(st.args)); break;
    case K_TABLES: hipLaunchKernelGGL(k_tables, st.blocks, kTablesThreads, 0, s, *reinterpret_cast<const TableAtt*>(st.args)); break;
    default: break;
  }
}
void launch_prep_steps_multi(int id, const void* items, const uint2* block_info, const uint32_t* item_blocks, uint32_t total_blocks, hipStream_t s) {
  switch (id) {
    case K_RANS_PREP: hipLaunchKernelGGL(k_rans_prep_multi, total_blocks, 256, 0, s, static_cast<const RansPrepArgs*>(items), block_info, item_blocks); break;
    case K_BATCH_FLAGS: hipLaunchKernelGGL(k_batch_flags_multi, total_blocks, 256, 0, s, static_cast<const BatchFlagsArgs*>(items), block_info, item_blocks); break;
    case K_BITS_PREP: hipLaunchKernelGGL(k_bits_prep_multi, total_blocks, 256, 0, s, static_cast<const BitsPrepArgs*>(items), block_info, item_blocks); break;
    case K_ORIENT_PREP: hipLaunchKernelGGL(k_orient_prep_multi, total_blocks, 64, 0, s, static_cast<const OrientPrepArgs*>(items), block_info, item_blocks); break;
    case K_TABLES: hipLaunchKernelGGL(k_tables_multi, total_blocks, kTablesThreads, 0, s, static_cast<const TableAtt*>(items), block_info, item_blocks); break;
    default: break;
  }
}

// record prep: symbols / bits / orientation flags → coding records, then the per-batch frequency-1 flags of the rABS record streams
// levels: the table kernel (device form) 0, record prep 1, batch flags of the rABS record streams 2
void launch_tables(const TableAtt& a, hipStream_t s) { emit_prep(K_TABLES, 0, a, 1u, s); }
void launch_tables_group(const TableGroup& g, hipStream_t s) { if (g.count > 0) hipLaunchKernelGGL(k_tables_group, (uint32_t)g.count, kTablesThreads, 0, s, g); }
void launch_rans_prep(const void* sym, bool sym16, uint64_t n, const RansEntry* table, uint32_t bins, RansEntry* rec, uint32_t* batch_flags, hipStream_t s) {
  RansPrepArgs a{sym, table, rec, batch_flags, n, bins, sym16 ? 1u : 0u};
  emit_prep(K_RANS_PREP, 1, a, n ? grid256(n) : 0u, s);
}
void launch_bits_prep(const uint8_t* bits, uint64_t n, RansEntry e0, RansEntry e1, RansEntry* rec, hipStream_t s) {
  BitsPrepArgs a{bits, rec, n, e0, e1, nullptr};
  emit_prep(K_BITS_PREP, 1, a, n ? grid256(n) : 0u, s);
}
void launch_bits_prep_dev(const uint8_t* bits, uint64_t n, const RansEntry* entries, RansEntry* rec, hipStream_t s) {
  BitsPrepArgs a{bits, rec, n, RansEntry{}, RansEntry{}, entries};
  emit_prep(K_BITS_PREP, 1, a, n ? grid256(n) : 0u, s);
}
void launch_orient_prep(const uint8_t* orient, uint32_t n, const uint32_t* chunk_info, RansEntry e0, RansEntry e1, RansEntry* rec, hipStream_t s) {
  OrientPrepArgs a{orient, chunk_info, rec, e0, e1, n, 0u, nullptr, nullptr};
  emit_prep(K_ORIENT_PREP, 1, a, (n + 4095u) / 4096u, s);
}
void launch_orient_prep_dev(const uint8_t* orient, uint32_t n, const uint32_t* chunk_info, const RansEntry* entries, RansEntry* rec, hipStream_t s) {
  OrientPrepArgs a{orient, chunk_info, rec, RansEntry{}, RansEntry{}, n, 0u, entries, nullptr};
  emit_prep(K_ORIENT_PREP, 1, a, (n + 4095u) / 4096u, s);
}
void launch_orient_bits(const uint8_t* orient, uint32_t n, const uint32_t* chunk_info, uint8_t* bits_out, hipStream_t s) {
  OrientPrepArgs a{orient, chunk_info, nullptr, RansEntry{}, RansEntry{}, n, 0u, nullptr, bits_out};
  emit_prep(K_ORIENT_PREP, 1, a, (n + 4095u) / 4096u, s);
}
void launch_batch_flags(const RansEntry* rec, uint64_t n, const uint32_t* n_dev, uint32_t* batch_flags, hipStream_t s) {
  BatchFlagsArgs a{rec, batch_flags, n, n_dev};
  emit_prep(K_BATCH_FLAGS, 2, a, n ? grid256(n) : 0u, s);
}
void launch_pack_streams(const ChainDesc* descs_dev, uint32_t n_streams, PackEntry* table, uint8_t* arena, hipStream_t s) {
  if (!n_streams) return;
  hipLaunchKernelGGL(k_pack_offsets, 1, 1024, 0, s, descs_dev, n_streams, table);
  hipLaunchKernelGGL(k_pack_copy, dim3(n_streams, kPackSplit), 256, 0, s, descs_dev, table, arena);
}
void launch_copy_items(const CopyItem* items_dev, uint32_t n_items, uint8_t* arena, hipStream_t s) {
  if (n_items) hipLaunchKernelGGL(k_copy_items, dim3(n_items, kPackSplit), 256, 0, s, items_dev, arena);
}
void launch_clear_items(const CopyItem* items_dev, uint32_t n_items, hipStream_t s) {
  if (n_items) hipLaunchKernelGGL(k_clear_items, n_items, 256, 0, s, items_dev);
}
void launch_clear_ranges(const ClearRanges& r, hipStream_t s) {
  if (!r.count) return;
  uint64_t longest = 0;
  for (uint32_t k = 0; k < r.count; ++k) longest = std::max(longest, r.bytes[k]);
  hipLaunchKernelGGL(k_clear_ranges, dim3(std::min(grid256(longest >> 2), 512u), r.count), 256, 0, s, r);
}
void launch_scatter_items(const CopyItem* items_dev, uint32_t n_items, const uint8_t* arena, hipStream_t s) {
  if (n_items) hipLaunchKernelGGL(k_scatter_items, n_items, 256, 0, s, items_dev, arena);
}
uint32_t chain_grid(uint32_t n_streams) {
  const uint32_t cap = dbg().chain_grid ? dbg().chain_grid : 256u;
  return n_streams < cap ? n_streams : cap;
}
// Which form a launch takes.  Dense: 4 pairs per CU, ≈ 20 ns per step (a walker shares its SIMD with an emitter).  Sparse: 2 pairs per CU,
// ≈ 16.2 ns per step.  A launch lasts about max(longest stream, all steps / walkers) steps — the sparse form wins while the longest
// stream dominates (single meshes, batches of a few hundred), the dense one when the sum does.
bool chain_launch_sparse(uint64_t longest_steps, uint64_t total_steps, uint32_t n_streams) {
  if (dbg_on(DMI_DBG_CHAIN_DENSE)) return false;
  const uint32_t g = chain_grid(n_streams);
  const double dense = 20.0 * (double)std::max<uint64_t>(longest_steps, total_steps / (4ull * g));
  const double sparse = 16.2 * (double)std::max<uint64_t>(longest_steps, total_steps / (2ull * g));
  return sparse <= dense;
}
void launch_chains(const ChainDesc* descs_dev, const uint32_t* order_dev, uint32_t n_streams, uint32_t* next_stream_dev, bool sparse, hipStream_t s) {
  if (!n_streams) return;
  (void)hipMemsetAsync(next_stream_dev, 0, sizeof(uint32_t), s);
  if (sparse) hipLaunchKernelGGL((k_chains<DMI_WALKER_PARK, 2u>), chain_grid(n_streams), 512, 0, s, descs_dev, order_dev, n_streams, next_stream_dev);
  else hipLaunchKernelGGL((k_chains<1u, 4u>), chain_grid(n_streams), 512, 0, s, descs_dev, order_dev, n_streams, next_stream_dev);
}

}  // namespace dmi
