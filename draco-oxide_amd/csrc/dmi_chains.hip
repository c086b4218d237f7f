// dmi_chains.hip — the serial entropy coders (encode/entropy/rans.rs:33-68 rANS, :91-128 rABS) for gfx950.
//
// The rANS/rABS state recurrence is ONE dependency chain per stream (a single non-interleaved
// stream per attribute), so a stream is owned by one wavefront and the chain runs on the SCALAR unit:
//   * k_rans_prep (data-parallel, all CUs): symbol → 16-byte coding record {M, b, D, c} in coding
//     (reverse) order, so the chain never touches the symbol table;
//   * k_chains: per 64 symbols the wave (a) vector-loads the 64 records (for byte emission),
//     (b) walks the 64 steps on SGPRs with s_load'ed records — per step: exact x/f by multiply-high,
//     renormalisation shift from the quotient's bit length, state update — parking each pre-renorm
//     state in its lane, (c) emits the renormalisation bytes of all 64 steps at wavefront prefix-sum
//     offsets.
//
// Exact division (x < 2^30, 1 ≤ f ≤ 2^20):  e = floor(log2 f)
//     f not a power of two:  M = ceil(2^(32+e) / f) ∈ (2^31, 2^32),  x / f = mulhi(x, M) >> e
//        (M·f = 2^(32+e) + ε with ε < f < 2^(e+1), and x·ε < 2^(31+e) < 2^(32+e) ⇒ exact)
//     f = 2^e, e ≥ 1:        M = 2^31,                                x / f = mulhi(x, M) >> (e-1)
//     f = 1:                 flagged; the batch takes the generic (divide) loop.
// Renormalisation: `while x ≥ f·2^10 { x >>= 8 }`  ⇔  k = (bitlen(x/f) - 3) >> 3  bytes
//     (x ≥ f·2^m ⇔ floor(x/f) ≥ 2^m, and floor(floor(x/2^8)/f) = floor(x/f) >> 8).
#include "dmi_device.hpp"

namespace dmi {
namespace {

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d, 64); if (lane >= (uint32_t)d) v += t; }
  return v;
}
__device__ __forceinline__ uint32_t div_magic(uint32_t f) { return f <= 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / f); }
// exact x / f with magic = floor(2^32/f): the estimate is q or q-1 for any 32-bit x
__device__ __forceinline__ void divmod_magic(uint32_t x, uint32_t f, uint32_t magic, uint32_t& q, uint32_t& r) {
  q = __umulhi(x, magic);
  r = x - q * f;
  if (r >= f) { ++q; r -= f; }
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

// symbol → coding record, in coding order (the stream is fed in reverse: symbol_coding.rs:161-163)
__global__ __launch_bounds__(256) void k_rans_prep(const uint32_t* __restrict__ sym, uint64_t n, const RansEntry* __restrict__ table,
                                                   RansEntry* __restrict__ rec) {
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (uint64_t)gridDim.x * 256) rec[t] = table[sym[n - 1 - t]];
}

struct Rec8 { RansEntry r[8]; };
typedef const Rec8 __attribute__((address_space(4))) * const_rec8_t;
__device__ __forceinline__ void load_rec8(Rec8& dst, const_rec8_t src) {
#pragma unroll
  for (int k = 0; k < 8; ++k) { dst.r[k].m = src->r[k].m; dst.r[k].b = src->r[k].b; dst.r[k].d = src->r[k].d; dst.r[k].c = src->r[k].c; }
}

#define DMI_RANS_STEP(R, J)                                                                     \
  {                                                                                              \
    const uint32_t q0 = __umulhi(x, (R).m) >> ((R).b & 31u);                                     \
    const uint32_t sh = (29u - (uint32_t)__builtin_clz(q0)) & 0x18u;                             \
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(parked) : "s"(x), "i"(J));                  \
    x = (x >> sh) + (q0 >> sh) * (R).d + (R).c;                                                  \
  }

__device__ void rans_chain(const ChainDesc& d, uint32_t lane) {
  const uint32_t P = d.precision;
  const uint64_t n = d.n;
  const RansEntry* __restrict__ rec = d.table;   // coding records, coding order
  uint32_t x = 4u << P;
  uint64_t pos = 0;
  uint32_t err = 0;
  for (uint64_t base = 0; base < n; base += 64) {
    const uint32_t cnt = (uint32_t)min((uint64_t)64, n - base);
    // (a) every lane fetches its step's record (freq = 2^P - d for the emission phase)
    RansEntry mine{0u, 0u, 0u, 0u};
    if (lane < cnt) mine = rec[base + lane];
    const bool has_f1 = __ballot((mine.b >> 8) != 0) != 0ull;   // a symbol with frequency 1 in this batch
    uint32_t parked = 0;
    x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    if (cnt == 64 && !has_f1) {
      // (b) 64 chain steps on the scalar unit, records fetched with scalar loads
      // constant address space ⇒ uniform loads become s_load_dwordx4/x8/x16 (the records were written
      // by the preceding k_rans_prep launch, so the scalar cache cannot hold stale lines)
      // Records are fetched 8 at a time (2 × s_load_dwordx16), one group ahead of the steps that use them.
      const_rec8_t g = (const_rec8_t)(uintptr_t)(rec + base);
      Rec8 cur;
      load_rec8(cur, g);
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) {
        Rec8 nxt = cur;
        if (gi + 1 < 8) load_rec8(nxt, g + gi + 1);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) DMI_RANS_STEP(cur.r[s8], gi * 8 + s8)
        cur = nxt;
      }
    } else {
      for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t dj = rl(mine.d, j), cj = rl(mine.c, j);
        const uint32_t f = (1u << P) - dj;
        const uint32_t q0 = x / f;
        const uint32_t sh = (29u - (uint32_t)__builtin_clz(q0)) & 0x18u;
        if (lane == j) parked = x;
        x = (x >> sh) + (q0 >> sh) * dj + cj;
      }
    }
    // (c) bytes of the 64 steps: lane j re-derives its byte count from (parked state, frequency)
    uint32_t my_k = 0;
    if (lane < cnt) {
      const uint32_t f = (1u << P) - mine.d;
      const uint64_t thr = (uint64_t)f << 10;   // ((L >> P) * f) << 8 with L = 4 << P
      uint64_t xx = parked;
      while (xx >= thr) { xx >>= 8; ++my_k; }
    }
    const uint32_t incl = wave_inclusive_scan(my_k, lane);
    const uint32_t total = rl(incl, 63);
    if (pos + total > d.cap) { err = 2; break; }
    const uint64_t at = pos + (incl - my_k);
    for (uint32_t t = 0; t < my_k; ++t) d.out[at + t] = (uint8_t)(parked >> (8 * t));
    pos += total;
  }
  if (lane == 0) {
    if (!err) pos += flush_state(x - (4u << P), d.out, pos, d.cap, err);
    d.out_len[0] = (uint32_t)pos;
    d.out_len[1] = err;
  }
}

// rABS with precision 8, L = 4096, single-`if` renormalisation (Q21).
// kind 1: bits[i] ∈ {0,1}, coded in forward order (normal flips, Q9).
// kind 2: orientation flags {0 none, 1 false, 2 true}; the coded bit of entry j is (o[j] == o[j+1]) with
//         o[len] = true, fed in forward order (mesh_prediction_for_texture_coordinates.rs:241-256, Q10).
__device__ void rabs_chain(const ChainDesc& d, uint32_t lane) {
  const uint32_t p0 = d.p0, f1 = 256u - p0;
  const uint32_t m0 = div_magic(p0), m1 = div_magic(f1);
  uint32_t x = 4096u;
  uint64_t pos = 0;
  uint32_t err = 0;
  uint32_t pending = 2;   // kind 2: orientation of the last valid entry still waiting for its successor
  const uint64_t n = d.n;
  for (uint64_t base = 0; base <= n && !err; base += 64) {
    const uint64_t idx = base + lane;
    uint32_t fl = 0;
    if (idx < n) fl = d.bits[idx];
    unsigned long long todo;      // lanes that contribute one coded bit, in lane order
    uint32_t bitv;                // the bit each such lane codes
    if (d.kind == 1) {
      todo = __ballot(idx < n);
      bitv = fl;
      if (base >= n) break;
    } else {
      const unsigned long long valid = __ballot(fl != 0);
      const unsigned long long ones = __ballot(fl == 2);
      const bool tail = base + 64 > n;   // last batch: flush the pending entry against `true`
      // lane L (valid) codes the bit of the PREVIOUS valid entry: (prev == mine)
      const unsigned long long below = valid & ((1ull << lane) - 1ull);
      uint32_t prev;
      if (below) { const int pl = 63 - __clzll(below); prev = (uint32_t)((ones >> pl) & 1ull); } else prev = pending;
      const uint32_t mine = (fl == 2);
      const bool codes = (fl != 0) && (prev != 2);
      todo = __ballot(codes);
      bitv = (prev == mine) ? 1u : 0u;
      if (valid) { const int ll = 63 - __clzll(valid); pending = (uint32_t)((ones >> ll) & 1ull); }
      if (tail) {
        // one extra coded bit for the final pending entry, issued by the first lane past the data
        const uint32_t extra_lane = (uint32_t)(n - base);   // 0..63
        if (pending != 2 && extra_lane < 64) {
          if (lane == extra_lane) bitv = (pending == 1u) ? 1u : 0u;
          todo |= (1ull << extra_lane);
        }
      }
    }
    uint32_t my_x = 0, my_k = 0, step = 0;
    unsigned long long rem = todo;
    while (rem) {
      const uint32_t j = (uint32_t)(__ffsll((long long)rem) - 1);
      rem &= rem - 1;
      const uint32_t bit = rl(bitv, j);
      const uint32_t f = bit ? f1 : p0, m = bit ? m1 : m0;
      uint32_t xr = x, k = 0;
      if (xr >= ((16u * f) << 8)) { xr >>= 8; k = 1; }
      if (lane == step) { my_x = x; my_k = k; }
      ++step;
      uint32_t q, r;
      divmod_magic(xr, f, m, q, r);
      x = (q << 8) + r + (bit ? 0u : f1);
    }
    const uint32_t incl = wave_inclusive_scan(my_k, lane);
    const uint32_t total = rl(incl, 63);
    if (pos + total > d.cap) { err = 2; break; }
    if (my_k) d.out[pos + (incl - my_k)] = (uint8_t)my_x;
    pos += total;
    if (d.kind == 2 && base + 64 > n) break;
  }
  if (lane == 0) {
    if (!err) pos += flush_state(x - 4096u, d.out, pos, d.cap, err);
    d.out_len[0] = (uint32_t)pos;
    d.out_len[1] = err;
  }
}

__global__ __launch_bounds__(64) void k_chains(const ChainDesc* __restrict__ descs) {
  const ChainDesc d = descs[blockIdx.x];
  if (d.kind == 0) rans_chain(d, threadIdx.x); else rabs_chain(d, threadIdx.x);
}

}  // namespace

void launch_rans_prep(const uint32_t* sym, uint64_t n, const RansEntry* table, RansEntry* rec, hipStream_t s) {
  if (!n) return;
  uint64_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_rans_prep, (uint32_t)g, 256, 0, s, sym, n, table, rec);
}

void launch_chains(const ChainDesc* descs_dev, uint32_t n_streams, hipStream_t s) {
  if (n_streams) hipLaunchKernelGGL(k_chains, n_streams, 64, 0, s, descs_dev);
}

}  // namespace dmi
