// dmi_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for the draco attribute-encoding
// hot path.  Integer / bit-twiddling work bounded by HBM bandwidth: no MFMA.  Build with
// -ffp-contract=off so every f32 operation rounds separately like the Rust reference.
//
// Reference arithmetic each kernel reproduces (paths relative to draco-oxide/src/):
//   k_value_ranges / _final            encode/attribute/portabilization/quantization_coordinate_wise.rs:24-68; prediction_transform/geom.rs:45
//   k_seq_quantize                     ...quantization_coordinate_wise.rs:70-91; octahedral_quantization.rs:49-64 + geom.rs:40-91,137-157;
//                                      to_bits.rs:41-49; attribute_encoder.rs:332-338 (sequence order); wrapped_difference.rs:36-52 (min/max)
//   k_build_fans                       (layout) core/corner_table/mod.rs swing_left/right around every coded vertex, once per job
//   k_predict_fused<pos,nrm,uv>        shared/attribute/prediction_scheme/mesh_parallelogram_prediction.rs:186-237 + wrapped_difference.rs:54-99,
//                                      .../mesh_normal_prediction.rs:22-44,75-144 + prediction_transform/oct_orthogonal.rs:23-85,
//                                      .../mesh_prediction_for_texture_coordinates.rs:32-81,107-219
//   k_pred_parallelogram_wrapped       the parallelogram predictor alone (custom attributes, attributes on seam tables)
//   k_pred_texcoord_wrapped            the texture-coordinate predictor alone (UV seams)
//   k_pred_delta_difference            .../delta_prediction.rs:56-71 + prediction_transform/difference.rs:26-34
//   k_orient_summary                   mesh_prediction_for_texture_coordinates.rs:224-235 (bit / transition counts)
//   k_histogram                        encode/entropy/symbol_coding.rs:149-157
//   (the serial rANS/rABS coders live in dmi_chains.hip)
#include "dmi_device.hpp"
#include <cstdlib>
#include <cstring>
#include <vector>

namespace dmi {
namespace {

constexpr int kBlock = 256;
constexpr uint32_t kNoneD = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t cnext(uint32_t c) { return (c % 3u == 2u) ? c - 2u : c + 1u; }
__device__ __forceinline__ uint32_t cprev(uint32_t c) { return (c % 3u == 0u) ? c + 2u : c - 1u; }

// Every kernel body is a __device__ function of (its argument block, blk_, nblk_) = (block index, block count) WITHIN ITS
// WORK ITEM: DMI_KERNEL generates the plain kernel (one item: blockIdx.x / gridDim.x) and the multi-item kernel that serves
// the same phase of many jobs in one launch (block_info[blockIdx.x] = {item, block within the item}).
// XCD-aware sweep of a sequence-ordered range.  Blocks are dealt round-robin over the 8 XCDs (block b and
// b+8 share an L2), and Edgebreaker-order neighbours of entry i sit one "ring" (a few thousand entries)
// before/after i.  Each XCD therefore owns one contiguous eighth of the chunk list and its blocks sweep it in
// step, so ring neighbours are fetched once per L2 instead of once per XCD.  (Placement affects speed only.)
#define DMI_FOR_SEQUENCE(I, N)                                                                           \
  for (uint32_t nch_ = ((N) + kBlock - 1) / kBlock, per_ = (nch_ + 7u) / 8u, xcd_ = blk_ & 7u,      \
                end_ = min(nch_, (xcd_ + 1u) * per_), ch_ = xcd_ * per_ + (blk_ >> 3);             \
       ch_ < end_; ch_ += (nblk_ >> 3))                                                              \
    for (uint32_t I = ch_ * kBlock + threadIdx.x; I < (N); I = (N))

// Same sweep in tiles of K·kBlock entries (each thread handles K entries of a tile, kBlock apart).
#define DMI_FOR_TILES(BASE, N, K)                                                                                   \
  for (uint32_t nch_ = ((N) + kBlock * (K) - 1) / (kBlock * (K)), per_ = (nch_ + 7u) / 8u, xcd_ = blk_ & 7u,   \
                end_ = min(nch_, (xcd_ + 1u) * per_), ch_ = xcd_ * per_ + (blk_ >> 3);                       \
       ch_ < end_; ch_ += (nblk_ >> 3))                                                                        \
    for (uint32_t BASE = ch_ * kBlock * (K), once_ = 1; once_; once_ = 0)

// Rust `as` casts: saturating, NaN → 0.
__device__ __forceinline__ int32_t f32_to_i32_sat(float f) {
  int32_t r;   // v_cvt_i32_f32 truncates, saturates and turns NaN into 0: the cast's semantics in one instruction
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}
__device__ __forceinline__ int64_t f32_to_i64_sat(float f) {
  if (f != f) return 0;
  if (f >= 9223372036854775808.0f) return 9223372036854775807ll;
  if (f <= -9223372036854775808.0f) return (-9223372036854775807ll - 1);
  return (int64_t)f;
}
__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
__device__ __forceinline__ int64_t wadd64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
__device__ __forceinline__ int64_t wsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
__device__ __forceinline__ int64_t wmul64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
__device__ __forceinline__ int64_t wabs64(int64_t a) { return a < 0 ? (int64_t)(0ull - (uint64_t)a) : a; }
__device__ __forceinline__ int64_t wdiv64(int64_t a, int64_t b) { return (b == -1) ? (int64_t)(0ull - (uint64_t)a) : a / b; }
__device__ __forceinline__ uint32_t zigzag(int32_t v) {   // utils/mod.rs:152-158
  return v >= 0 ? ((uint32_t)v << 1) : ((((uint32_t)(-(v + 1))) << 1) + 1u);
}

// entry of corner c in a per-corner table: dense arrays (stride 3: the corner itself) or face records (stride 8: see FusedArgs::face_stride)
__device__ __forceinline__ size_t cidx(uint32_t c, uint32_t face_stride) { return face_stride == 8u ? (size_t)(c / 3u) * 8u + c % 3u : (size_t)c; }

// ---- layouts of quantized values and symbols (QFmt, sym16: dmi_device.hpp) ----
__device__ __forceinline__ uint64_t pack_p64(int32_t x, int32_t y, int32_t z) { return (uint64_t)(uint32_t)x | ((uint64_t)(uint32_t)y << 21) | ((uint64_t)(uint32_t)z << 42); }
__device__ __forceinline__ void unpack_p64(uint64_t v, int32_t (&out)[3]) {
  out[0] = (int32_t)((uint32_t)v & 0x1FFFFFu); out[1] = (int32_t)((uint32_t)(v >> 21) & 0x1FFFFFu); out[2] = (int32_t)((uint32_t)(v >> 42) & 0x1FFFFFu);
}
// quantized position of sequence entry r (zeros for a missing rank) from a QF_I32 or QF_P64 array
__device__ __forceinline__ void load_pos_fmt(const void* __restrict__ q, int fmt, uint32_t r, int32_t (&out)[3]) {
  if (r == 0xFFFFFFFFu) { out[0] = 0; out[1] = 0; out[2] = 0; }
  else if (fmt == QF_P64) unpack_p64(static_cast<const uint64_t*>(q)[r], out);
  else { const int32_t* p = static_cast<const int32_t*>(q) + (size_t)r * 3; out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; }
}
// symbol stores of the predictors: nontemporal by default (A/B: -DDMI_SYM_TEMPORAL keeps them in L2 / MALL for the histogram that follows)
#ifdef DMI_SYM_TEMPORAL
#define DMI_SYM_STORE(v, p) (*(p) = (v))
#else
#define DMI_SYM_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif
__device__ __forceinline__ void store_sym(void* __restrict__ sym, bool s16, size_t idx, uint32_t v) {
  if (s16) DMI_SYM_STORE((uint16_t)v, static_cast<uint16_t*>(sym) + idx);
  else DMI_SYM_STORE(v, static_cast<uint32_t*>(sym) + idx);
}
__device__ __forceinline__ uint32_t load_sym(const void* __restrict__ sym, bool s16, uint64_t idx) {
  return s16 ? (uint32_t)static_cast<const uint16_t*>(sym)[idx] : static_cast<const uint32_t*>(sym)[idx];
}

// ------------------------------------------------------------------------------------------------
// min / max of f32 components, both seeded with +0.0 (quirk Q1).  Reductions use the reference's
// own comparisons (`<`, `>`): a partial can never be -0.0 or NaN, so the tree order is irrelevant.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void block_reduce_minmax(float (&mn)[N], float (&mx)[N], float* out) {
  __shared__ float sh[2 * N * (kBlock / 64)];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float a = __shfl_down(mn[k], off, 64);
      float b = __shfl_down(mx[k], off, 64);
      if (a < mn[k]) mn[k] = a;
      if (b > mx[k]) mx[k] = b;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) { sh[wave * 2 * N + k] = mn[k]; sh[wave * 2 * N + N + k] = mx[k]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        float a = sh[w * 2 * N + k], b = sh[w * 2 * N + N + k];
        if (a < mn[k]) mn[k] = a;
        if (b > mx[k]) mx[k] = b;
      }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) { out[k] = mn[k]; out[N + k] = mx[k]; }
  }
}

// ------------------------------------------------------------------------------------------------
// Stage 1, every attribute of a job in two launches (same-address atomics from a full-chip grid serialise at
// ≈11 ns each — scripts/probes/atomics_probe.hip — so every reduction here goes through per-block partials).
//   k_value_ranges        block → (attribute, slice): f32 min/max partials (coordinate-wise attributes) or the
//                         zero-length-normal check (geom.rs:45 assert) as a per-block flag
//   k_value_ranges_final  one block per attribute: partials → meta (min[N], shared range Q2, max[N]); seeds the
//                         16 scratch words: [0..1] joint i32 min/max = {INT_MAX, INT_MIN}, [4] zero-normal flag, rest 0
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void range_slice(const RangeAtt& a, uint32_t block, float* out) {
  float mn[N], mx[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { mn[k] = 0.0f; mx[k] = 0.0f; }
  // DMI_RANGE_UNROLL values per thread and round, all loads issued before the compares
#ifndef DMI_RANGE_UNROLL
#define DMI_RANGE_UNROLL 2   // (4: 33.8 µs, 8: 39.3, 2: 32.2 for the 160 MB of the 10M-triangle workload on 3 × 1024 blocks)
#endif
  const uint32_t stride = a.blocks * kBlock;
  for (uint32_t v0 = block * kBlock + threadIdx.x; v0 < a.n; v0 += DMI_RANGE_UNROLL * stride) {
    float x[DMI_RANGE_UNROLL][N];
#pragma unroll
    for (int u = 0; u < DMI_RANGE_UNROLL; ++u) {
      const uint32_t v = v0 + u * stride;
#pragma unroll
      for (int k = 0; k < N; ++k) x[u][k] = (v < a.n) ? a.raw[(size_t)v * N + k] : 0.0f;   // 0.0 is the seed: neutral
    }
#pragma unroll
    for (int u = 0; u < DMI_RANGE_UNROLL; ++u) {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        if (x[u][k] < mn[k]) mn[k] = x[u][k];
        if (x[u][k] > mx[k]) mx[k] = x[u][k];
      }
    }
  }
  block_reduce_minmax<N>(mn, mx, out);
}
__device__ __forceinline__ void k_value_ranges_body(const RangeArgs& args, const uint32_t blk_, const uint32_t nblk_) {
  int ai = 0;
  while (ai + 1 < args.count && blk_ >= args.a[ai + 1].first_block) ++ai;
  const RangeAtt& a = args.a[ai];
  const uint32_t block = blk_ - a.first_block;
  if (a.kind == 0) {
    float* out = a.partials + (size_t)block * 2 * a.N;
    switch (a.N) {
      case 1: range_slice<1>(a, block, out); break;
      case 2: range_slice<2>(a, block, out); break;
      case 3: range_slice<3>(a, block, out); break;
      default: range_slice<4>(a, block, out); break;
    }
  } else if (a.kind == 1) {
    bool bad = false;
    for (uint32_t v = block * kBlock + threadIdx.x; v < a.n; v += a.blocks * kBlock)
      bad |= (a.raw[(size_t)v * 3] == 0.0f && a.raw[(size_t)v * 3 + 1] == 0.0f && a.raw[(size_t)v * 3 + 2] == 0.0f);
    const int any = __syncthreads_or(bad ? 1 : 0);
    if (threadIdx.x == 0) reinterpret_cast<uint32_t*>(a.partials)[block] = any ? 1u : 0u;
  }
}

template <int N>
__device__ __forceinline__ void range_final(const RangeAtt& a) {
  float mn[N], mx[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { mn[k] = 0.0f; mx[k] = 0.0f; }
  for (uint32_t b = threadIdx.x; b < a.blocks; b += kBlock) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float lo = a.partials[(size_t)b * 2 * N + k], hi = a.partials[(size_t)b * 2 * N + N + k];
      if (lo < mn[k]) mn[k] = lo;
      if (hi > mx[k]) mx[k] = hi;
    }
  }
  __shared__ float res[2 * N];
  block_reduce_minmax<N>(mn, mx, res);
  __syncthreads();
  if (threadIdx.x == 0) {
    float delta_max = 0.0f;   // one shared range (Q2)
#pragma unroll
    for (int k = 0; k < N; ++k) { const float d = res[N + k] - res[k]; if (d > delta_max) delta_max = d; }
#pragma unroll
    for (int k = 0; k < N; ++k) { a.meta[k] = res[k]; a.meta[N + 1 + k] = res[N + k]; }
    a.meta[N] = delta_max;
  }
}
__device__ __forceinline__ void k_value_ranges_final_body(const RangeArgs& args, const uint32_t blk_, const uint32_t nblk_) {
  const RangeAtt a = args.a[blk_];
  if (threadIdx.x < 16) a.small[threadIdx.x] = threadIdx.x == 0 ? 0x7FFFFFFFu : (threadIdx.x == 1 ? 0x80000000u : 0u);
  // the rest of the attribute's slab slot (ranges, histogram, orientation summaries) starts every encode at zero
  for (size_t w = threadIdx.x; w < a.zero_words; w += kBlock) a.zero[w] = 0u;
  __syncthreads();   // the seeding stores are performed before any flag / range store below
  if (a.kind == 0) {
    switch (a.N) {
      case 1: range_final<1>(a); break;
      case 2: range_final<2>(a); break;
      case 3: range_final<3>(a); break;
      default: range_final<4>(a); break;
    }
  } else if (a.kind == 1) {
    bool bad = false;
    for (uint32_t b = threadIdx.x; b < a.blocks; b += kBlock) bad |= reinterpret_cast<const uint32_t*>(a.partials)[b] != 0u;
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) a.small[4] = 1u;   // same value from every writer
  }
}

// q = trunc(((v - min) / range) * (2^bits - 1) + 0.5), range == 0 skips the divide (Q3); every f32 operation
// rounds separately (no FMA contraction), the cast is Rust's `as i64 as i32`.
__device__ __forceinline__ int32_t quant_coord(float v, float mn, float range, float maxq) {
  const float diff = v - mn;
  const float normalized = (range == 0.0f) ? diff : diff / range;
  const float quantized = normalized * maxq;
  const float h = quantized + 0.5f;
  if (fabsf(h) < 2147483648.0f) return (int32_t)h;   // (the i64 conversion is ≈ 20 instructions; its low word is this whenever it fits)
  return (int32_t)f32_to_i64_sat(h);
}

// geom.rs:40-91 (f32 path; Q5: the fold uses the pre-fold u and v; Q6: no normalisation)
__device__ __forceinline__ void oct_transform(float x, float y, float z, float& u, float& v) {
  const float abs_sum = fabsf(x) + fabsf(y) + fabsf(z);
  u = y / abs_sum;
  v = z / abs_sum;
  if (x < 0.0f) {
    const float uo = (u < 0.0f) ? fabsf(v) - 1.0f : 1.0f - fabsf(v);
    const float vo = (v < 0.0f) ? fabsf(u) - 1.0f : 1.0f - fabsf(u);
    u = uo;
    v = vo;
  }
}
// geom.rs:137-157 (Q7)
__device__ __forceinline__ void oct_faithful(int32_t u, int32_t v, int32_t& x, int32_t& y) {
  x = u; y = v;
  if ((u == 0 && v == 0) || (u == 255 && v == 0) || (u == 0 && v == 255)) { x = 255; y = 255; }
  else if (u == 0 && v > 127) y = 127 - (v - 127);
  else if (u == 255 && v < 127) y = 127 + (127 - v);
  else if (v == 255 && u < 127) x = 127 + (127 - u);
  else if (v == 0 && u > 127) x = 127 - (u - 127);
}
__device__ __forceinline__ void oct_quantize(float x, float y, float z, int32_t& qx, int32_t& qy) {
  float u, v;
  oct_transform(x, y, z, u, v);
  const float a = (u + 1.0f) * 127.0f;   // (1 << 8-1) - 1 = 127 (Q4)
  const float b = (v + 1.0f) * 127.0f;
  oct_faithful(f32_to_i32_sat(a), f32_to_i32_sat(b), qx, qy);
}

// ------------------------------------------------------------------------------------------------
// Portabilization in coding order.  Entry i of the sequence is quantized straight into qs[i]:
//   qs[i] = portabilize(raw[value(point(seq[i]))])        (attribute_encoder.rs:332-338 reads exactly these)
// The index chain seq → corner_to_point → point_to_value is composed once at job creation (s2p / s2v).
// so no value-ordered quantized array is ever materialised.  Every attribute coded on the same corner table
// is served by one launch (they share seq / corner_to_point traffic); the joint i32 min/max of each attribute
// (wrapped_difference.rs:36-52, Q16) is reduced on the fly.  All corners of one (attribute-)vertex carry the
// same value, so qs[c2r[c]] == attribute.get(point_idx(c)) for every corner c of the table.
// ------------------------------------------------------------------------------------------------
// Each thread owns kTile entries of a tile (entry t of the tile at base + t·kBlock + threadIdx.x) and issues all of
// their gathers before touching any of them: the pass is latency-bound otherwise (one 12-byte gather in flight per
// lane is ≈3 TB/s by Little's law at HBM latency).
// KT (entries per thread): 1 in the single-mesh launch, 2 in the batch launch — measured at 8 waves per SIMD: 10M-triangle mesh 85.0 µs at 1
// against 89.9 at 2 (98.1 at 3: spills); 1024-mesh batch 457 µs at 2 against 494 at 1.
#ifndef DMI_KTILE_SINGLE
#define DMI_KTILE_SINGLE 1
#endif
#ifndef DMI_KTILE_MULTI
#define DMI_KTILE_MULTI 2
#endif
template <int N, int KT> struct RawTile { float v[KT][N]; };
// issue: the KT gathers of one attribute (value indices first when the attribute has its own point → value map)
template <int N, int KT>
__device__ __forceinline__ void gather_tile(const QuantAtt& a, const uint32_t (&p)[KT], const uint32_t (&d)[KT], uint32_t base, uint32_t n, RawTile<N, KT>& r) {
  uint32_t v[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const uint32_t i = base + t * kBlock + threadIdx.x;
    v[t] = (a.s2v && i < n) ? a.s2v[d[t]] : p[t];
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
#pragma unroll
    for (int k = 0; k < N; ++k) r.v[t][k] = a.raw[(size_t)v[t] * N + k];   // (entries past n read value 0: harmless)
  }
}
// retire: quantize and store the tile
template <int N, int KT>
__device__ __forceinline__ void finish_tile(const QuantAtt& a, const RawTile<N, KT>& r, const uint32_t (&d)[KT], uint32_t base, uint32_t n, int32_t& mn, int32_t& mx) {
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const uint32_t slot = base + t * kBlock + threadIdx.x;
    const uint32_t i = d[t];   // sequence index this slot is written to (= slot unless the pass runs tile-sorted)
    int32_t out[N];
    int nq = N;
    if (a.kind == 0) {   // coordinate-wise (meta: min[N], range)
#pragma unroll
      for (int k = 0; k < N; ++k) out[k] = quant_coord(r.v[t][k], a.meta[k], a.meta[N], a.maxq);
    } else if (a.kind == 1) {   // octahedral (N == 3 → 2 components)
      if constexpr (N == 3) { int32_t u, w; oct_quantize(r.v[t][0], r.v[t][1], r.v[t][2], u, w); out[0] = u; out[1] = w; nq = 2; }
    } else {   // ToBits: the 4-byte values reinterpreted as i32
#pragma unroll
      for (int k = 0; k < N; ++k) out[k] = __float_as_int(r.v[t][k]);
    }
    if (slot < n) {
      for (int k = 0; k < nq; ++k) { mn = min(mn, out[k]); mx = max(mx, out[k]); }
      if (a.fmt == QF_P64) { if (N == 3) static_cast<uint64_t*>(a.qs)[i] = pack_p64(out[0], out[1], out[2]); }
      else if (a.fmt == QF_B16) static_cast<uint16_t*>(a.qs)[i] = (uint16_t)((uint32_t)out[0] | ((uint32_t)out[1] << 8));
      else if (a.fmt == QF_H32) static_cast<uint32_t*>(a.qs)[i] = (uint32_t)out[0] | ((uint32_t)out[1] << 16);
      else { int32_t* q = static_cast<int32_t*>(a.qs); for (int k = 0; k < nq; ++k) q[(size_t)i * nq + k] = out[k]; }
    }
  }
}
template <int N, int KT>
__device__ __forceinline__ void quantize_tile(const QuantAtt& a, const uint32_t (&p)[KT], const uint32_t (&d)[KT], uint32_t base, uint32_t n, int32_t& mn, int32_t& mx) {
  RawTile<N, KT> r;
  gather_tile<N, KT>(a, p, d, base, n, r);
  finish_tile<N, KT>(a, r, d, base, n, mn, mx);
}
// The common attribute sets of one table issue EVERY attribute's gathers before the first value is touched: a store between two
// attributes' gathers orders them (the pointers may alias as far as the compiler knows), and the tile would pay one memory latency
// per attribute instead of one.
template <int N0, int N1, int KT>
__device__ __forceinline__ void quantize_tiles2(const QuantArgs& q, const uint32_t (&p)[KT], const uint32_t (&d)[KT], uint32_t base, uint32_t n, int32_t (&mn)[kMaxGather], int32_t (&mx)[kMaxGather]) {
  RawTile<N0, KT> r0; RawTile<N1, KT> r1;
  gather_tile<N0, KT>(q.a[0], p, d, base, n, r0); gather_tile<N1, KT>(q.a[1], p, d, base, n, r1);
  finish_tile<N0, KT>(q.a[0], r0, d, base, n, mn[0], mx[0]); finish_tile<N1, KT>(q.a[1], r1, d, base, n, mn[1], mx[1]);
}
template <int N0, int N1, int N2, int KT>
__device__ __forceinline__ void quantize_tiles3(const QuantArgs& q, const uint32_t (&p)[KT], const uint32_t (&d)[KT], uint32_t base, uint32_t n, int32_t (&mn)[kMaxGather], int32_t (&mx)[kMaxGather]) {
  RawTile<N0, KT> r0; RawTile<N1, KT> r1; RawTile<N2, KT> r2;
  gather_tile<N0, KT>(q.a[0], p, d, base, n, r0); gather_tile<N1, KT>(q.a[1], p, d, base, n, r1); gather_tile<N2, KT>(q.a[2], p, d, base, n, r2);
  finish_tile<N0, KT>(q.a[0], r0, d, base, n, mn[0], mx[0]); finish_tile<N1, KT>(q.a[1], r1, d, base, n, mn[1], mx[1]); finish_tile<N2, KT>(q.a[2], r2, d, base, n, mn[2], mx[2]);
}
template <int KT>
__device__ __forceinline__ void k_seq_quantize_body(const SeqQuantArgs& sq, const uint32_t blk_, const uint32_t nblk_) {
  const uint32_t* __restrict__ s2p = sq.s2p;
  const uint32_t n = sq.n;
  const QuantArgs& args = sq.q;
  int32_t mn[kMaxGather], mx[kMaxGather];
#pragma unroll
  for (int a = 0; a < kMaxGather; ++a) { mn[a] = 2147483647; mx[a] = (-2147483647 - 1); }
  const int sig = args.count * 1000 + (args.count > 0 ? args.a[0].N * 100 : 0) + (args.count > 1 ? args.a[1].N * 10 : 0) + (args.count > 2 ? args.a[2].N : 0);
  DMI_FOR_TILES(base, n, KT) {
    uint32_t p[KT], d[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) { const uint32_t i = base + t * kBlock + threadIdx.x; p[t] = i < n ? (s2p ? s2p[i] : i) : 0u; d[t] = (sq.dest && i < n) ? sq.dest[i] : i; }   // (s2p null: value order — the early stage of a whole-mesh call)
    if (sig == 3332) { quantize_tiles3<3, 3, 2, KT>(args, p, d, base, n, mn, mx); continue; }   // position, normal, texture coordinate
    if (sig == 3323) { quantize_tiles3<3, 2, 3, KT>(args, p, d, base, n, mn, mx); continue; }
    if (sig == 2330) { quantize_tiles2<3, 3, KT>(args, p, d, base, n, mn, mx); continue; }
    if (sig == 2320) { quantize_tiles2<3, 2, KT>(args, p, d, base, n, mn, mx); continue; }
#pragma unroll
    for (int a = 0; a < kMaxGather; ++a) {
      if (a >= args.count) break;
      switch (args.a[a].N) {
        case 1: quantize_tile<1, KT>(args.a[a], p, d, base, n, mn[a], mx[a]); break;
        case 2: quantize_tile<2, KT>(args.a[a], p, d, base, n, mn[a], mx[a]); break;
        case 3: quantize_tile<3, KT>(args.a[a], p, d, base, n, mn[a], mx[a]); break;
        default: quantize_tile<4, KT>(args.a[a], p, d, base, n, mn[a], mx[a]); break;
      }
    }
  }
  __shared__ int32_t red[kMaxGather][2][kBlock / 64];
#pragma unroll
  for (int a = 0; a < kMaxGather; ++a) {
    if (a >= args.count) break;
    int32_t lo = mn[a], hi = mx[a];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); }
    if ((threadIdx.x & 63) == 0) { red[a][0][threadIdx.x >> 6] = lo; red[a][1][threadIdx.x >> 6] = hi; }
  }
  __syncthreads();
  if (threadIdx.x < (uint32_t)args.count) {   // one partial per block and attribute; k_i32_minmax_final folds them
    int32_t lo = red[threadIdx.x][0][0], hi = red[threadIdx.x][1][0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) { lo = min(lo, red[threadIdx.x][0][w]); hi = max(hi, red[threadIdx.x][1][w]); }
    args.a[threadIdx.x].ipartials[2 * blk_] = lo;
    args.a[threadIdx.x].ipartials[2 * blk_ + 1] = hi;
  }
}

// The early stage of a whole-mesh call and the coding-order gather that is left of the quantize step (round 5).  A call whose values are in HBM when
// it starts (dmi_encode_mesh_device) waits ≈ 90 ms for the host's two serial walks before a sequence exists: value ranges and the quantization
// itself — in VALUE order, a streaming pass — run on a side stream meanwhile (early_quantize_issue, dmi_job.cpp).  For a mesh whose attributes are
// all per-point (no point → value map: every attribute is indexed by the point) the quantized position, normal and texture coordinate of a
// value go into ONE 16-byte record — u64 position (x | y << 21 | z << 42), u32 texture coordinate (u | v << 16), u16 octahedral normal — so that the
// pass, once the sequence is known, pays one 16-byte gather per entry instead of three gathers of 12 + 12 + 8 bytes and the quantizer's arithmetic
// (a wavefront-wide gather costs by the distinct lines it touches, not by the bytes it keeps: three attributes were three times the lines).
// The arithmetic is finish_tile's, call for call: quant_coord per coordinate, oct_quantize for the normal.
struct alignas(16) QuantRec { uint64_t pos; uint32_t uv; uint16_t nrm; uint16_t pad; };
// min / max of a block's share of per-block f32 partial pairs ([mn[N] | mx[N]] per producer block, k_value_ranges' layout), reduced over the block:
// res[0..N) = min, res[N..2N) = max, visible to every thread on return.  The reference's own comparisons (`<`, `>`), seeds +0.0 (Q1).
template <int N>
__device__ __forceinline__ void reduce_range_partials(const float* __restrict__ partials, uint32_t blocks, float* res) {
  float mn[N], mx[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { mn[k] = 0.0f; mx[k] = 0.0f; }
  for (uint32_t b = threadIdx.x; b < blocks; b += kBlock) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float lo = partials[(size_t)b * 2 * N + k], hi = partials[(size_t)b * 2 * N + N + k];
      if (lo < mn[k]) mn[k] = lo;
      if (hi > mx[k]) mx[k] = hi;
    }
  }
  block_reduce_minmax<N>(mn, mx, res);
  __syncthreads();
}
// One joint i32 min/max pair per block and attribute as well (wrapped_difference.rs:36-52 takes them over what the SEQUENCE holds: the sequence of a
// per-point attribute on the position's table holds every value exactly once — job_create_impl adopts the stage only then); the first block of the
// kernel that consumes the records folds them (k_seq_gather_rec).
// Round 6: no `_final` launches around it.  Every block folds the range partials of k_value_ranges itself (512 pairs per attribute out of L2: ≈ 1 µs beside
// 45 µs of streaming) — the shared range (Q2) is the same arithmetic in the same order as k_value_ranges_final's, and block 0 also writes what that
// kernel wrote into the attribute's slot (ranges, seeded scratch words, the zero-normal flag) — and keeps kRecPer values per thread in flight: every
// load of a tile is issued before the first value is used (the first form waited for each attribute's row in turn: 56 µs for 240 MB).
// (HAS_NRM / HAS_UV as template parameters: tested at run time, the attribute's loads sat in branches of their own and the compiler waited for each)
constexpr int kRecPer = 4;
template <bool HAS_NRM, bool HAS_UV>
__device__ __forceinline__ void value_quantize_rec_body(const ValueRecArgs& a) {
  __shared__ float res_pos[6], res_uv[4];
  reduce_range_partials<3>(a.pos_partials, a.range_blocks[0], res_pos);
  if (HAS_UV) reduce_range_partials<2>(a.uv_partials, a.range_blocks[2], res_uv);
  float pmn[3], prange = 0.0f, umn[2] = {0.0f, 0.0f}, urange = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { pmn[k] = res_pos[k]; const float d = res_pos[3 + k] - res_pos[k]; if (d > prange) prange = d; }   // one shared range (Q2)
  if (HAS_UV) {
#pragma unroll
    for (int k = 0; k < 2; ++k) { umn[k] = res_uv[k]; const float d = res_uv[2 + k] - res_uv[k]; if (d > urange) urange = d; }
  }
  if (blockIdx.x == 0) {   // the slots of the attributes: [small 16 words][meta 16 words] each, as k_value_ranges_final leaves them
    const uint32_t t = threadIdx.x;
    // (word 4 of the normal's slot — a zero-length normal seen, geom.rs:45 — is folded by the consumer's first block from this kernel's per-block flags:
    //  the normals are read HERE, once; the range pass read them a first time for that check alone until round 6 — 60 of its 160 MB)
    if (t < 96u) {
      const uint32_t k = t >> 5, w = t & 31u;
      uint32_t* slot = a.slot[k];
      if (slot) {
        uint32_t v = w == 0 ? 0x7FFFFFFFu : (w == 1 ? 0x80000000u : 0u);
        if (k == 0 && w >= 16u) { const uint32_t m = w - 16u; if (m < 3u) v = __float_as_uint(res_pos[m]); else if (m == 3u) v = __float_as_uint(prange); else if (m < 7u) v = __float_as_uint(res_pos[3 + (m - 4u)]); }
        if (k == 2 && w >= 16u) { const uint32_t m = w - 16u; if (m < 2u) v = __float_as_uint(res_uv[m]); else if (m == 2u) v = __float_as_uint(urange); else if (m < 5u) v = __float_as_uint(res_uv[2 + (m - 3u)]); }
        slot[w] = v;
      }
    }
  }
  int32_t mn[3], mx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { mn[k] = 2147483647; mx[k] = (-2147483647 - 1); }
  bool zero_normal = false;
  const uint32_t n = a.n;
  constexpr uint32_t kTileRec = kBlock * kRecPer;
  for (uint32_t base = blockIdx.x * kTileRec; base < n; base += gridDim.x * kTileRec) {
    float P[kRecPer][3], Q[kRecPer][3], U[kRecPer][2];
#pragma unroll
    for (int t = 0; t < kRecPer; ++t) {
      const uint32_t v0 = base + t * kBlock + threadIdx.x;
      const size_t v = v0 < n ? v0 : n - 1u;
#pragma unroll
      for (int k = 0; k < 3; ++k) P[t][k] = a.pos[3 * v + k];
      if (HAS_NRM) {
#pragma unroll
        for (int k = 0; k < 3; ++k) Q[t][k] = a.nrm[3 * v + k];
      }
      if (HAS_UV) { U[t][0] = a.uv[2 * v]; U[t][1] = a.uv[2 * v + 1]; }
    }
#pragma unroll
    for (int t = 0; t < kRecPer; ++t) {
      const uint32_t v = base + t * kBlock + threadIdx.x;
      QuantRec r{0ull, 0u, 0u, 0u};
      const int32_t q0 = quant_coord(P[t][0], pmn[0], prange, a.pos_maxq), q1 = quant_coord(P[t][1], pmn[1], prange, a.pos_maxq), q2 = quant_coord(P[t][2], pmn[2], prange, a.pos_maxq);
      r.pos = pack_p64(q0, q1, q2);
      int32_t u = 0, w = 0, qu = 0, qw = 0;
      if (HAS_NRM) { oct_quantize(Q[t][0], Q[t][1], Q[t][2], u, w); r.nrm = (uint16_t)((uint32_t)u | ((uint32_t)w << 8)); }
      if (HAS_UV) { qu = quant_coord(U[t][0], umn[0], urange, a.uv_maxq); qw = quant_coord(U[t][1], umn[1], urange, a.uv_maxq); r.uv = (uint32_t)qu | ((uint32_t)qw << 16); }
      if (v < n) {
        mn[0] = min(mn[0], min(q0, min(q1, q2))); mx[0] = max(mx[0], max(q0, max(q1, q2)));
        if (HAS_NRM) { mn[1] = min(mn[1], min(u, w)); mx[1] = max(mx[1], max(u, w)); zero_normal |= Q[t][0] == 0.0f && Q[t][1] == 0.0f && Q[t][2] == 0.0f; }
        if (HAS_UV) { mn[2] = min(mn[2], min(qu, qw)); mx[2] = max(mx[2], max(qu, qw)); }
        static_cast<QuantRec*>(a.rec)[v] = r;
      }
    }
  }
  if (HAS_NRM) {
    const int any = __syncthreads_or(zero_normal ? 1 : 0);
    if (threadIdx.x == 0) a.nrm_flags[blockIdx.x] = any ? 1u : 0u;
  }
  __shared__ int32_t red[3][2][kBlock / 64];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int32_t lo = mn[k], hi = mx[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); }
    if ((threadIdx.x & 63) == 0) { red[k][0][threadIdx.x >> 6] = lo; red[k][1][threadIdx.x >> 6] = hi; }
  }
  __syncthreads();
  if (threadIdx.x < 3u && a.ipartials[threadIdx.x]) {   // (0 position, 1 normal, 2 texture coordinate)
    int32_t lo = red[threadIdx.x][0][0], hi = red[threadIdx.x][1][0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) { lo = min(lo, red[threadIdx.x][0][w]); hi = max(hi, red[threadIdx.x][1][w]); }
    a.ipartials[threadIdx.x][2 * blockIdx.x] = lo;
    a.ipartials[threadIdx.x][2 * blockIdx.x + 1] = hi;
  }
}
__global__ __launch_bounds__(kBlock) void k_value_quantize_rec_pnu(const ValueRecArgs a) { value_quantize_rec_body<true, true>(a); }
__global__ __launch_bounds__(kBlock) void k_value_quantize_rec_pn(const ValueRecArgs a) { value_quantize_rec_body<true, false>(a); }
__global__ __launch_bounds__(kBlock) void k_value_quantize_rec_pu(const ValueRecArgs a) { value_quantize_rec_body<false, true>(a); }
// The first block of the early stage's consumer: the joint i32 min/max of every attribute (the per-block pairs of k_value_quantize_rec, folded here: no
// `_final` launch) into words 0–1 of the attribute's slot in the job's slab, the other 30 words of the stage's slot (ranges, seeded scratch words, the
// zero-normal flag) copied beside them — [small 64 B][meta 64 B] per attribute.  All kBlock threads of the block call it.
__device__ __forceinline__ void early_slots_block(const EarlySlots& e) {
  __shared__ int32_t red[2][kBlock / 64];
  for (int k = 0; k < 3; ++k) {
    if (!e.src[k] || !e.dst[k]) continue;   // (uniform)
    int32_t lo = 2147483647, hi = (-2147483647 - 1);
    if (e.ipartials[k]) for (uint32_t b = threadIdx.x; b < e.ipartial_blocks; b += kBlock) { lo = min(lo, e.ipartials[k][2 * b]); hi = max(hi, e.ipartials[k][2 * b + 1]); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); }
    __syncthreads();   // (red of the previous attribute has been read)
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int w = 1; w < kBlock / 64; ++w) { lo = min(lo, red[0][w]); hi = max(hi, red[1][w]); }
      e.dst[k][0] = (uint32_t)lo; e.dst[k][1] = (uint32_t)hi;
    } else if (threadIdx.x >= 2u && threadIdx.x < 32u) e.dst[k][threadIdx.x] = e.src[k][threadIdx.x];
  }
  if (e.nrm_flags && e.dst[1]) {   // (uniform) a zero-length normal seen by any block of the quantizer (geom.rs:45) → word 4 of the normal's slot
    bool bad = false;
    for (uint32_t b = threadIdx.x; b < e.ipartial_blocks; b += kBlock) bad |= e.nrm_flags[b] != 0u;
    const int any = __syncthreads_or(bad ? 1 : 0);   // (also orders this store behind the copy of word 4 above)
    if (threadIdx.x == 4u) e.dst[1][4] = any ? 1u : 0u;
  }
}
// qs_*[i] = the fields of rec[s2p[i]].  Four entries per thread, every gather issued before the first is used; tiles are dealt to blocks like every
// pass over a sequence (an XCD's blocks take one contiguous eighth of it, so that the rings its gathers revisit stay in ITS L2).
constexpr int kGatherPer = 4;
__global__ __launch_bounds__(kBlock) void k_seq_gather_rec(const GatherRecArgs g) {
  const uint32_t n = g.n;
  const uint32_t blk_ = blockIdx.x, nblk_ = gridDim.x;
  if (blk_ == 0) early_slots_block(g.slots);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* __restrict__ rec = static_cast<const u32x4*>(g.rec);
  DMI_FOR_TILES(base, n, kGatherPer) {
    uint32_t p[kGatherPer];
#pragma unroll
    for (int t = 0; t < kGatherPer; ++t) { const uint32_t i = base + t * kBlock + threadIdx.x; p[t] = i < n ? g.s2p[i] : 0u; }
    u32x4 r[kGatherPer];
#pragma unroll
    for (int t = 0; t < kGatherPer; ++t) r[t] = rec[p[t]];
#pragma unroll
    for (int t = 0; t < kGatherPer; ++t) {
      const uint32_t i = base + t * kBlock + threadIdx.x;
      if (i >= n) continue;
      g.qs_pos[i] = (uint64_t)r[t].x | ((uint64_t)r[t].y << 32);
      if (g.qs_uv) g.qs_uv[i] = r[t].z;
      if (g.qs_nrm) g.qs_nrm[i] = (uint16_t)(r[t].w & 0xFFFFu);
    }
  }
}

// Joint i32 min/max (wrapped_difference.rs:36-52, Q16) of every attribute: one block per attribute.
__device__ __forceinline__ void k_i32_minmax_final_body(const MinMaxArgs& args, const uint32_t blk_, const uint32_t nblk_) {
  const MinMaxAtt& a = args.a[blk_];
  int32_t lo = 2147483647, hi = (-2147483647 - 1);
  // two partial pairs per 16-byte load, eight loads in flight per thread: the 8192 partials of a large mesh are two rounds, not 32 dependent ones
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  const i32x4* __restrict__ p4 = reinterpret_cast<const i32x4*>(a.ipartials);
  const uint32_t n4 = a.blocks >> 1;
  for (uint32_t b = threadIdx.x; b < n4; b += kBlock * 8u) {
    i32x4 v[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) { const uint32_t at = b + u * kBlock; v[u] = at < n4 ? p4[at] : i32x4{2147483647, (-2147483647 - 1), 2147483647, (-2147483647 - 1)}; }
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) { lo = min(lo, min(v[u].x, v[u].z)); hi = max(hi, max(v[u].y, v[u].w)); }
  }
  if ((a.blocks & 1u) && threadIdx.x == 0) { lo = min(lo, a.ipartials[2 * (a.blocks - 1)]); hi = max(hi, a.ipartials[2 * (a.blocks - 1) + 1]); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); }
  __shared__ int32_t red[2][kBlock / 64];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) { lo = min(lo, red[0][w]); hi = max(hi, red[1][w]); }
    a.minmax[0] = lo;
    a.minmax[1] = hi;
  }
}

// WrappedDifference::squeeze parameters from the joint min/max (wrapped_difference.rs:62-69, Q16)
struct WrapParams { int32_t mn, mx, max_diff, max_corr, min_corr; };
__device__ __forceinline__ WrapParams wrap_params(const int32_t* minmax) {
  WrapParams w;
  w.mn = minmax[0]; w.mx = minmax[1];
  w.max_diff = wadd(1, wsub(w.mx, w.mn));
  w.max_corr = w.max_diff / 2;
  w.min_corr = (int32_t)(0u - (uint32_t)w.max_corr);
  if ((w.max_diff & 1) == 0) w.max_corr = wsub(w.max_corr, 1);
  return w;
}
__device__ __forceinline__ uint32_t wrap_symbol(int32_t orig, int32_t pred, const WrapParams& w) {
  pred = pred < w.mn ? w.mn : (pred > w.mx ? w.mx : pred);
  const int32_t val = wsub(orig, pred);
  int32_t corr = val;
  if (val > w.max_corr) corr = wsub(val, w.max_diff);
  else if (val < w.min_corr) corr = wadd(val, w.max_diff);
  return zigzag(corr);
}

template <int N>
__device__ __forceinline__ void k_pred_parallelogram_wrapped_body(const ParArgs& pa, const uint32_t blk_, const uint32_t nblk_) {
  const uint32_t* __restrict__ seq = pa.seq;
  const uint32_t n = pa.n;
  const uint32_t* __restrict__ c2r = pa.c2r;
  const uint32_t* __restrict__ opp = pa.opp;
  const int32_t* __restrict__ qs = pa.qs;
  const int32_t* __restrict__ minmax = pa.minmax;
  void* __restrict__ sym = pa.sym;
  const bool s16 = pa.sym16 != 0u;
  const WrapParams w = wrap_params(minmax);
  DMI_FOR_SEQUENCE(i, n) {
    const uint32_t c = seq[i];
    const uint32_t o = opp[c];
    int32_t pred[N];
    bool have = false;
    if (o != kNoneD) {
      const uint32_t ro = c2r[o], rn = c2r[cnext(c)], rp = c2r[cprev(c)];
      if (ro < i && rn < i && rp < i) {
        have = true;
#pragma unroll
        for (int k = 0; k < N; ++k) pred[k] = wsub(wadd(qs[(size_t)rn * N + k], qs[(size_t)rp * N + k]), qs[(size_t)ro * N + k]);
      }
    }
    if (!have) {
      // value of the previously coded vertex (Q15): qs[i-1]; zero for the very first entry
#pragma unroll
      for (int k = 0; k < N; ++k) pred[k] = (i > 0) ? qs[(size_t)(i - 1) * N + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) store_sym(sym, s16, (size_t)i * N + k, wrap_symbol(qs[(size_t)i * N + k], pred[k], w));
  }
}

__device__ __forceinline__ void k_pred_delta_difference_body(const DeltaArgs& da, const uint32_t blk_, const uint32_t nblk_) {
  const uint64_t n_comp = da.n_comp;
  const int N = da.N;
  const int32_t* __restrict__ qs = da.qs;
  void* __restrict__ sym = da.sym;
  const bool s16 = da.sym16 != 0;
  for (uint64_t e = (uint64_t)blk_ * kBlock + threadIdx.x; e < n_comp; e += (uint64_t)nblk_ * kBlock) {
    const int32_t pred = (e >= (uint64_t)N) ? qs[e - N] : 0;
    store_sym(sym, s16, e, zigzag(wsub(qs[e], pred)));
  }
}

// oct_orthogonal.rs:23-74
__device__ __forceinline__ int32_t isgn(int32_t v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }
__device__ __forceinline__ int32_t iabs(int32_t v) { return v < 0 ? (int32_t)(0u - (uint32_t)v) : v; }
__device__ __forceinline__ void oct_orthogonal(int32_t o0, int32_t o1, int32_t p0, int32_t p1, uint32_t& s0, uint32_t& s1) {
  // (every value here is an octahedral coordinate or a difference of two: |v| ≤ 2^9, so the wrapping i32 products of the reference are
  //  exact on the full-rate 24-bit multiplier, and a product with a sign is a select)
  const int32_t one = 127;
  p0 = wsub(p0, one); p1 = wsub(p1, one); o0 = wsub(o0, one); o1 = wsub(o1, one);
  if (wadd(iabs(p0), iabs(p1)) > one) {
    const int32_t pa = p0, qs = -isgn(__mul24(p0, p1));
    p0 = wadd(__mul24(qs, p1), __mul24(isgn(pa), one));
    p1 = wadd(__mul24(qs, pa), __mul24(isgn(p1), one));
    const int32_t oa = o0, qo = -isgn(__mul24(o0, o1));
    o0 = wadd(__mul24(qo, o1), __mul24(isgn(oa), one));
    o1 = wadd(__mul24(qo, oa), __mul24(isgn(o1), one));
  }
  // The reference turns both points by quarter turns (x, y) → (−y, x) until the prediction lies in {p0 < 0, p1 <= 0}: the number of
  // turns only depends on the prediction's quadrant — none there, one for {p0 <= 0, p1 > 0}, two for {p0 > 0, p1 >= 0}, three for
  // {p0 >= 0, p1 < 0} (the four sets tile the plane without the origin, which is not turned at all).
  const int turns = (p0 == 0 && p1 == 0) ? 0 : ((p0 < 0 && p1 <= 0) ? 0 : ((p0 <= 0 && p1 > 0) ? 1 : ((p0 > 0 && p1 >= 0) ? 2 : 3)));
  const bool swap = (turns & 1) != 0, neg0 = turns == 1 || turns == 2, neg1 = turns == 2 || turns == 3;
  {
    const int32_t a0 = swap ? p1 : p0, a1 = swap ? p0 : p1, b0 = swap ? o1 : o0, b1 = swap ? o0 : o1;
    p0 = neg0 ? (int32_t)(0u - (uint32_t)a0) : a0; p1 = neg1 ? (int32_t)(0u - (uint32_t)a1) : a1;
    o0 = neg0 ? (int32_t)(0u - (uint32_t)b0) : b0; o1 = neg1 ? (int32_t)(0u - (uint32_t)b1) : b1;
  }
  int32_t c0 = wsub(o0, p0), c1 = wsub(o1, p1);
  if (c0 < 0) c0 = wadd(c0, 255);
  if (c1 < 0) c1 = wadd(c1, 255);
  s0 = (uint32_t)c0; s1 = (uint32_t)c1;
}

// Exact i64 `a / d` (truncating) for the texture-coordinate predictor.  The software 64-bit divide costs ~100
// instructions; when |a| and d fit 52 bits the f64 quotient is within one of the answer and one remainder
// check fixes it.  Anything larger takes the generic divide.
__device__ __forceinline__ int64_t div_exact(int64_t a, int64_t d) {
  const int64_t lim = 1ll << 52;
  if (d > 0 && d < lim && a > -lim && a < lim) {
    int64_t q = (int64_t)((double)a / (double)d);
    const int64_t r = a - q * d;
    if (a >= 0) { if (r < 0) --q; else if (r >= d) ++q; }
    else { if (r > 0) ++q; else if (r <= -d) --q; }
    return q;
  }
  return wdiv64(a, d);
}
// `a > i64::MAX / b` without the divide: for a ≥ 0, b > 0 it is a·b > i64::MAX (128-bit product)
__device__ __forceinline__ bool exceeds_max_over(int64_t a, int64_t b) {
  if (a < 0 || b <= 0) return a > wdiv64(9223372036854775807ll, b);
  const uint64_t hi = __umul64hi((uint64_t)a, (uint64_t)b), lo = (uint64_t)a * (uint64_t)b;
  return hi != 0ull || lo > 9223372036854775807ull;
}

// mesh_prediction_for_texture_coordinates.rs:32-48
__device__ __forceinline__ uint64_t udiv_exact(uint64_t a, uint64_t d) {
  const uint64_t lim = 1ull << 52;
  if (a < lim && d < lim) {
    uint64_t q = (uint64_t)((double)a / (double)d);
    const int64_t r = (int64_t)a - (int64_t)(q * d);
    if (r < 0) --q; else if (r >= (int64_t)d) ++q;
    return q;
  }
  return a / d;
}
__device__ __forceinline__ uint64_t int_sqrt(uint64_t value) {
  if (value == 0) return 0;
  uint64_t act = value, sq = 1;
  while (act >= 2) { sq *= 2; act /= 4; }
  sq = (sq + udiv_exact(value, sq)) / 2;
  while (sq * sq > value) sq = (sq + udiv_exact(value, sq)) / 2;
  return sq;
}

// General form (any i32 operands, wrapping i64 arithmetic exactly as the reference's release build): cu = the entry's
// own UV, nu / pu = UVs of the next / previous corner's vertex (both coded), cp / np / pp = the three quantised
// positions (get_position_for_vertex :22-30).  Returns false when the fallback must be used.
// (all operands and results travel by value: references to the callers' arrays would force them into scratch memory on every
// entry — 64 bytes of private-segment stores per vertex, ≈ 0.3 GB of HBM writes on the 10M workload — even though this
// out-of-line form almost never runs)
struct TexPred { int32_t pred0, pred1; uint32_t oflag; uint32_t ok; };
__device__ __noinline__ TexPred texcoord_predict_general_v(int32_t cu0, int32_t cu1, int32_t nuv0, int32_t nuv1, int32_t puv0, int32_t puv1, int32_t cp0, int32_t cp1,
                                                          int32_t cp2, int32_t np0, int32_t np1, int32_t np2, int32_t pp0, int32_t pp1, int32_t pp2);
__device__ __forceinline__ bool texcoord_predict_general(const int32_t (&cu)[2], const int32_t (&nuv)[2], const int32_t (&puv)[2], const int32_t (&cpi)[3],
                                                        const int32_t (&npi)[3], const int32_t (&ppi)[3], int32_t& pred0, int32_t& pred1, uint8_t& oflag) {
  const TexPred r = texcoord_predict_general_v(cu[0], cu[1], nuv[0], nuv[1], puv[0], puv[1], cpi[0], cpi[1], cpi[2], npi[0], npi[1], npi[2], ppi[0], ppi[1], ppi[2]);
  if (r.ok) { pred0 = r.pred0; pred1 = r.pred1; if (r.oflag) oflag = (uint8_t)r.oflag; }
  return r.ok != 0;
}
__device__ __forceinline__ bool texcoord_predict_general_impl(const int32_t (&cu)[2], const int32_t (&nuv)[2], const int32_t (&puv)[2], const int32_t (&cpi)[3],
                                                 const int32_t (&npi)[3], const int32_t (&ppi)[3], int32_t& pred0, int32_t& pred1, uint8_t& oflag) {
  const int64_t nu0 = nuv[0], nu1 = nuv[1], pu0 = puv[0], pu1 = puv[1];
  if (nu0 == pu0 && nu1 == pu1) { pred0 = (int32_t)pu0; pred1 = (int32_t)pu1; return true; }   // degenerate: identical neighbour UVs
  const int64_t cp[3] = {cpi[0], cpi[1], cpi[2]}, np[3] = {npi[0], npi[1], npi[2]}, pp[3] = {ppi[0], ppi[1], ppi[2]};
  const int64_t pn0 = wsub64(pp[0], np[0]), pn1 = wsub64(pp[1], np[1]), pn2 = wsub64(pp[2], np[2]);
  const uint64_t pn2sq = (uint64_t)wadd64(wadd64(wmul64(pn0, pn0), wmul64(pn1, pn1)), wmul64(pn2, pn2));
  if (pn2sq == 0) return false;
  const int64_t cn0 = wsub64(cp[0], np[0]), cn1 = wsub64(cp[1], np[1]), cn2 = wsub64(cp[2], np[2]);
  const int64_t cn_dot_pn = wadd64(wadd64(wmul64(pn0, cn0), wmul64(pn1, cn1)), wmul64(pn2, cn2));
  const int64_t pnu0 = wsub64(pu0, nu0), pnu1 = wsub64(pu1, nu1);
  const int64_t n_uv_absmax = max(wabs64(nu0), wabs64(nu1));
  const int64_t pn_uv_absmax = max(wabs64(pnu0), wabs64(pnu1));
  const int64_t pn_absmax = max(max(wabs64(pn0), wabs64(pn1)), wabs64(pn2));
  const bool overflow = exceeds_max_over(n_uv_absmax, (int64_t)pn2sq) || exceeds_max_over(wabs64(cn_dot_pn), pn_uv_absmax) ||
                        exceeds_max_over(wabs64(cn_dot_pn), pn_absmax);
  if (overflow) return false;
  const int64_t xu0 = wadd64(wmul64(nu0, (int64_t)pn2sq), wmul64(pnu0, cn_dot_pn));
  const int64_t xu1 = wadd64(wmul64(nu1, (int64_t)pn2sq), wmul64(pnu1, cn_dot_pn));
  const int64_t xp0 = wadd64(np[0], div_exact(wmul64(pn0, cn_dot_pn), (int64_t)pn2sq));
  const int64_t xp1 = wadd64(np[1], div_exact(wmul64(pn1, cn_dot_pn), (int64_t)pn2sq));
  const int64_t xp2 = wadd64(np[2], div_exact(wmul64(pn2, cn_dot_pn), (int64_t)pn2sq));
  const int64_t cx0 = wsub64(cp[0], xp0), cx1 = wsub64(cp[1], xp1), cx2 = wsub64(cp[2], xp2);
  const uint64_t cx2sq = (uint64_t)wadd64(wadd64(wmul64(cx0, cx0), wmul64(cx1, cx1)), wmul64(cx2, cx2));
  const uint64_t nrm = int_sqrt(cx2sq * pn2sq);
  const int64_t cxu0 = wmul64(pnu1, (int64_t)nrm), cxu1 = wmul64((int64_t)(0ull - (uint64_t)pnu0), (int64_t)nrm);
  const int64_t a0 = div_exact(wadd64(xu0, cxu0), (int64_t)pn2sq), a1 = div_exact(wadd64(xu1, cxu1), (int64_t)pn2sq);
  const int64_t b0 = div_exact(wsub64(xu0, cxu0), (int64_t)pn2sq), b1 = div_exact(wsub64(xu1, cxu1), (int64_t)pn2sq);
  const int64_t ea0 = wsub64(cu[0], a0), ea1 = wsub64(cu[1], a1), eb0 = wsub64(cu[0], b0), eb1 = wsub64(cu[1], b1);
  const int64_t da = wadd64(wmul64(ea0, ea0), wmul64(ea1, ea1)), db = wadd64(wmul64(eb0, eb0), wmul64(eb1, eb1));
  if (da < db) { oflag = 2; pred0 = (int32_t)a0; pred1 = (int32_t)a1; }
  else { oflag = 1; pred0 = (int32_t)b0; pred1 = (int32_t)b1; }
  return true;
}

__device__ __noinline__ TexPred texcoord_predict_general_v(int32_t cu0, int32_t cu1, int32_t nuv0, int32_t nuv1, int32_t puv0, int32_t puv1, int32_t cp0, int32_t cp1,
                                                          int32_t cp2, int32_t np0, int32_t np1, int32_t np2, int32_t pp0, int32_t pp1, int32_t pp2) {
  const int32_t cu[2] = {cu0, cu1}, nuv[2] = {nuv0, nuv1}, puv[2] = {puv0, puv1}, cp[3] = {cp0, cp1, cp2}, np[3] = {np0, np1, np2}, pp[3] = {pp0, pp1, pp2};
  TexPred r{0, 0, 0u, 0u};
  uint8_t oflag = 0;
  r.ok = texcoord_predict_general_impl(cu, nuv, puv, cp, np, pp, r.pred0, r.pred1, oflag) ? 1u : 0u;
  r.oflag = oflag;
  return r;
}

__device__ __forceinline__ bool fits31(int64_t v) { return v > -(1ll << 31) && v < (1ll << 31); }

// One entry of mesh_prediction_for_texture_coordinates.rs:107-219 on already fetched operands: cu = the entry's
// own UV, nu / pu = UVs of the next / previous corner's vertex (both coded), cp / np / pp = the three quantised
// positions (get_position_for_vertex :22-30).  Returns false when the fallback must be used.
// Quantised values are < 2^30, so first-level differences fit i32 and their products are single 32×32→64 multiplies;
// when |pn|², cn·pn fit 31 bits (neighbouring vertices: always, in practice) the reference's three overflow guards are
// provably false.  Every shortcut is exact integer arithmetic on values that cannot wrap, so the results equal the
// general form's; operands outside these ranges take texcoord_predict_general.
// (floor(sqrt(value)) is what the reference's Newton iteration (:32-48) converges to from any start: every integer Newton step from
//  s > 0 lands at or above it and the loop then descends onto it; the f64 tier below forms it directly.)
// trunc(a / d) for integers held in doubles, |a| < 2^53, 0 < d < 2^31, inv ≈ 1 / d: the product a·inv is within one of the quotient,
// the remainder a − q·d is exact in one fma (it is an integer below 2^33), and one step fixes the quotient — ten full-rate
// instructions where the i64 form pays two emulated i64 ↔ f64 conversions and a 64-bit multiply.
__device__ __forceinline__ double div_trunc_f64(double a, double d, double inv) {
  double q = trunc(a * inv);
  const double r = fma(-q, d, a);
  if (a >= 0.0) { if (r < 0.0) q -= 1.0; else if (r >= d) q += 1.0; }
  else { if (r > 0.0) q += 1.0; else if (r <= -d) q -= 1.0; }
  return q;
}
// The projection in exact f64 arithmetic, for the operand sizes of ordinary meshes: texture coordinates below 2^20, |pn|² and cn·pn
// below 2^31 (as the integer tier below), the foot of the perpendicular within 2^11 of the corner, residuals below 2^26.  Every product
// and sum formed here is an integer of magnitude < 2^53 — exactly representable — so the results are the i64 form's.  Returns 0: not
// applicable (take the integer tiers), 1: predicted, 2: the reference's own fallback (|pn|² = 0 is handled by the caller's integer tier).
__device__ __forceinline__ int texcoord_predict_f64(const int32_t (&cu)[2], const int32_t (&nuv)[2], const int32_t (&puv)[2], const int32_t (&cp)[3],
                                                    const int32_t (&np)[3], const int32_t (&pn)[3], int32_t d32, int32_t c32, int32_t& pred0, int32_t& pred1, uint8_t& oflag) {
  const uint32_t lim20 = 1u << 20;
  if (!((uint32_t)nuv[0] < lim20 && (uint32_t)nuv[1] < lim20 && (uint32_t)puv[0] < lim20 && (uint32_t)puv[1] < lim20 && (uint32_t)cu[0] < lim20 && (uint32_t)cu[1] < lim20)) return 0;
  const double d = (double)d32, c = (double)c32;
  const double inv = 1.0 / d;
  double cx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double foot = div_trunc_f64((double)pn[k] * c, d, inv);   // |pn_k|·|c| < 2^16·2^31
    cx[k] = (double)cp[k] - ((double)np[k] + foot);
    if (!(fabs(cx[k]) < 2048.0)) return 0;
  }
  const double cx2sq = fma(cx[0], cx[0], fma(cx[1], cx[1], cx[2] * cx[2]));   // < 3·2^22
  if (!(cx2sq < 2097152.0)) return 0;                                            // (the integer tier's bound: cx2sq·|pn|² < 2^52)
  const double v = cx2sq * d;
  double nrm = trunc(sqrt(v));                                                   // floor(sqrt(v)) up to one unit; the squares below are exact (nrm < 2^26)
  if (nrm * nrm > v) nrm -= 1.0;
  else if ((nrm + 1.0) * (nrm + 1.0) <= v) nrm += 1.0;
  const double pnu0 = (double)(puv[0] - nuv[0]), pnu1 = (double)(puv[1] - nuv[1]);
  const double xu0 = fma((double)nuv[0], d, pnu0 * c), xu1 = fma((double)nuv[1], d, pnu1 * c);   // |·| < 2^51 + 2^51
  const double cxu0 = pnu1 * nrm, cxu1 = -(pnu0 * nrm);                          // < 2^46
  // the two candidates one after the other (fewer doubles alive at once); residuals below 2^26 keep the squared distances exact
  const double lim26 = 67108864.0;
  const double a0 = div_trunc_f64(xu0 + cxu0, d, inv), a1 = div_trunc_f64(xu1 + cxu1, d, inv);
  const double ea0 = (double)cu[0] - a0, ea1 = (double)cu[1] - a1;
  if (!(fabs(ea0) < lim26 && fabs(ea1) < lim26)) return 0;
  const double da = fma(ea0, ea0, ea1 * ea1);                                    // < 2^53
  const int32_t ia0 = (int32_t)a0, ia1 = (int32_t)a1;                           // (|a| < 2^27: the casts are exact)
  const double b0 = div_trunc_f64(xu0 - cxu0, d, inv), b1 = div_trunc_f64(xu1 - cxu1, d, inv);
  const double eb0 = (double)cu[0] - b0, eb1 = (double)cu[1] - b1;
  if (!(fabs(eb0) < lim26 && fabs(eb1) < lim26)) return 0;
  const double db = fma(eb0, eb0, eb1 * eb1);
  if (da < db) { oflag = 2; pred0 = ia0; pred1 = ia1; }
  else { oflag = 1; pred0 = (int32_t)b0; pred1 = (int32_t)b1; }
  return 1;
}

// One entry in the exact f64 tier (ordinary operand sizes).  1: predicted, 0: the reference's own fallback (|pn|² = 0), 2: operands outside
// the tier — the caller takes the general i64 form (texcoord_predict) or defers the entry to k_texcoord_fixup (the fused sweeps).
__device__ __forceinline__ int texcoord_try(const int32_t (&cu)[2], const int32_t (&nuv)[2], const int32_t (&puv)[2], const int32_t (&cp)[3],
                                            const int32_t (&np)[3], const int32_t (&pp)[3], int32_t& pred0, int32_t& pred1, uint8_t& oflag) {
  if (nuv[0] == puv[0] && nuv[1] == puv[1]) { pred0 = puv[0]; pred1 = puv[1]; return 1; }   // degenerate: identical neighbour UVs
#ifndef DMI_NO_F64_TEXCOORD
  bool small = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) small = small && (uint32_t)cp[k] < (1u << 30) && (uint32_t)np[k] < (1u << 30) && (uint32_t)pp[k] < (1u << 30);
  if (small) {
    const uint32_t lim15 = 1u << 15;   // |pn| components below 2^15: |pn|² < 3·2^30 — formed in uint32 (no signed wrap), admitted below 2^31 only
    const int32_t pn[3] = {pp[0] - np[0], pp[1] - np[1], pp[2] - np[2]};
    const int32_t cn[3] = {cp[0] - np[0], cp[1] - np[1], cp[2] - np[2]};
    if ((uint32_t)iabs(pn[0]) < lim15 && (uint32_t)iabs(pn[1]) < lim15 && (uint32_t)iabs(pn[2]) < lim15) {
      const uint32_t du = (uint32_t)(pn[0] * pn[0]) + (uint32_t)(pn[1] * pn[1]) + (uint32_t)(pn[2] * pn[2]);   // each square < 2^30
      if (du == 0) return 0;                                                   // mesh_prediction_for_texture_coordinates.rs: |pn|² = 0 → fallback
      const int32_t d32 = (int32_t)du;
      const int64_t cdp = (int64_t)pn[0] * cn[0] + (int64_t)pn[1] * cn[1] + (int64_t)pn[2] * cn[2];
      if (du < (1u << 31) && fits31(cdp) && texcoord_predict_f64(cu, nuv, puv, cp, np, pn, d32, (int32_t)cdp, pred0, pred1, oflag) == 1) return 1;
    }
  }
#endif
  return 2;
}
// One entry: the exact f64 tier in line (ordinary operand sizes), everything else through the out-of-line general form.
__device__ __forceinline__ bool texcoord_predict(const int32_t (&cu)[2], const int32_t (&nuv)[2], const int32_t (&puv)[2], const int32_t (&cp)[3],
                                                 const int32_t (&np)[3], const int32_t (&pp)[3], int32_t& pred0, int32_t& pred1, uint8_t& oflag) {
  const int st = texcoord_try(cu, nuv, puv, cp, np, pp, pred0, pred1, oflag);
  if (st != 2) return st == 1;
  return texcoord_predict_general(cu, nuv, puv, cp, np, pp, pred0, pred1, oflag);
}

constexpr int kTexTile = 2;   // the i64 projection is register-hungry: two entries per thread
__device__ __forceinline__ void k_pred_texcoord_wrapped_body(const TexArgs& ta, const uint32_t blk_, const uint32_t nblk_) {
  const uint32_t* __restrict__ seq = ta.seq;
  const uint32_t n = ta.n;
  const uint32_t* __restrict__ c2r = ta.c2r;
  const int32_t* __restrict__ qs = ta.qs;
  const uint32_t* __restrict__ c2r_pos = ta.c2r_pos;
  const void* __restrict__ qs_pos = ta.qs_pos;
  const int pos_fmt = ta.pos_fmt;
  const int32_t* __restrict__ minmax = ta.minmax;
  void* __restrict__ sym = ta.sym;
  const bool s16 = ta.sym16 != 0u;
  uint8_t* __restrict__ orient = ta.orient;
  const WrapParams w = wrap_params(minmax);
  const bool shared = (c2r_pos == c2r);   // both attributes on one table: the position ranks are i, rn, rp themselves
  DMI_FOR_TILES(base, n, kTexTile) {
    uint32_t i[kTexTile], c[kTexTile], rn[kTexTile], rp[kTexTile], qc[kTexTile], qn[kTexTile], qp[kTexTile];
#pragma unroll
    for (int t = 0; t < kTexTile; ++t) { i[t] = min(base + t * kBlock + threadIdx.x, n - 1u); c[t] = seq[i[t]]; }
#pragma unroll
    for (int t = 0; t < kTexTile; ++t) {
      const uint32_t nc = cnext(c[t]), pc = cprev(c[t]);
      rn[t] = c2r[nc]; rp[t] = c2r[pc];
      if (!shared) { qc[t] = c2r_pos[c[t]]; qn[t] = c2r_pos[nc]; qp[t] = c2r_pos[pc]; }
    }
    int32_t cu[kTexTile][2], nu[kTexTile][2], pu[kTexTile][2], lastv[kTexTile][2], cp[kTexTile][3], np[kTexTile][3], pp[kTexTile][3];
    bool both[kTexTile];
#pragma unroll
    for (int t = 0; t < kTexTile; ++t) {
      if (shared) { qc[t] = i[t]; qn[t] = rn[t]; qp[t] = rp[t]; }
      both[t] = rn[t] < i[t] && rp[t] < i[t];
      const uint32_t prev = i[t] > 0 ? i[t] - 1u : 0u;
      const size_t in_ = (size_t)(rn[t] < i[t] ? rn[t] : prev) * 2, ip_ = (size_t)(rp[t] < i[t] ? rp[t] : prev) * 2;
      cu[t][0] = qs[(size_t)i[t] * 2]; cu[t][1] = qs[(size_t)i[t] * 2 + 1];
      nu[t][0] = qs[in_]; nu[t][1] = qs[in_ + 1];
      pu[t][0] = qs[ip_]; pu[t][1] = qs[ip_ + 1];
      lastv[t][0] = qs[(size_t)prev * 2]; lastv[t][1] = qs[(size_t)prev * 2 + 1];
      // positions are only read when both neighbours are coded; a missing rank (never coded) reads as zero
      load_pos_fmt(qs_pos, pos_fmt, (both[t] && qc[t] != kNoneD) ? qc[t] : 0u, cp[t]);
      load_pos_fmt(qs_pos, pos_fmt, (both[t] && qn[t] != kNoneD) ? qn[t] : 0u, np[t]);
      load_pos_fmt(qs_pos, pos_fmt, (both[t] && qp[t] != kNoneD) ? qp[t] : 0u, pp[t]);
    }
#pragma unroll
    for (int t = 0; t < kTexTile; ++t) {
      if (base + t * kBlock + threadIdx.x >= n) continue;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (qc[t] == kNoneD) cp[t][d] = 0;
        if (qn[t] == kNoneD) np[t][d] = 0;
        if (qp[t] == kNoneD) pp[t][d] = 0;
      }
      int32_t pred0 = 0, pred1 = 0;
      uint8_t oflag = 0;
      bool done = false;
      if (both[t]) done = texcoord_predict(cu[t], nu[t], pu[t], cp[t], np[t], pp[t], pred0, pred1, oflag);
      if (!done) {   // fallback_predict :51-81 (the `prev` branch is intentionally absent)
        oflag = 0;
        if (rn[t] < i[t]) { pred0 = nu[t][0]; pred1 = nu[t][1]; }
        else if (i[t] > 0) { pred0 = lastv[t][0]; pred1 = lastv[t][1]; }
        else { pred0 = 0; pred1 = 0; }
      }
      orient[i[t]] = oflag;
      store_sym(sym, s16, (size_t)i[t] * 2, wrap_symbol(cu[t][0], pred0, w));
      store_sym(sym, s16, (size_t)i[t] * 2 + 1, wrap_symbol(cu[t][1], pred1, w));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// All predictors of a seam-free mesh in ONE sweep.  When the normal / texture-coordinate attributes are coded on
// the same corner table as their parent position attribute (no seams: the attribute tables alias the universal
// one), entry i of every attribute is the same vertex and the three predictors walk the same neighbourhood:
//   * the connectivity of the pass is the entry's fan row (see k_build_fans): fetched once instead of `opp`/`c2r` chases
//     per attribute;
//   * the quantised positions of the 1-ring feed the parallelogram, the texture-coordinate projection AND the
//     normal predictor, whose per-face cross products (mesh_normal_prediction.rs:22-44) are formed on the fly from
//     the ring positions — each fan face brings in exactly one new vertex — so no per-face normal array exists.
// Results are identical to the per-attribute kernels (k_pred_parallelogram_wrapped<3>, k_predict_fused<0,1,0>,
// k_pred_texcoord_wrapped): the tests run both paths against the oracle.
// ------------------------------------------------------------------------------------------------
// sum += cross(a - c, b - c): i32 wrapping terms, widened to i64 (wrapping sum).
// M24: the operands are differences of values below 2^21 (QF_P64 positions), so they fit 24 signed bits and the low 32 bits of their
// product — all the wrapping multiply keeps — come from the full-rate 24-bit multiplier (v_mul_i32_i24) instead of the quarter-rate
// 32-bit one: six multiplies per fan face are a fifth of the sweep's vector work.
template <bool M24>
__device__ __forceinline__ int32_t mul_w(int32_t a, int32_t b) { return M24 ? __mul24(a, b) : wmul(a, b); }
// sum += cross(ea, eb) of two edge vectors (already relative to the fan's centre): the body of add_face_normal
template <bool M24>
__device__ __forceinline__ void add_edge_cross(const int32_t (&ea)[3], const int32_t (&eb)[3], int64_t (&sum)[3]) {
  sum[0] = wadd64(sum[0], (int64_t)wsub(mul_w<M24>(ea[1], eb[2]), mul_w<M24>(ea[2], eb[1])));
  sum[1] = wadd64(sum[1], (int64_t)wsub(mul_w<M24>(ea[2], eb[0]), mul_w<M24>(ea[0], eb[2])));
  sum[2] = wadd64(sum[2], (int64_t)wsub(mul_w<M24>(ea[0], eb[1]), mul_w<M24>(ea[1], eb[0])));
}
template <bool M24>
__device__ __forceinline__ void add_face_normal(const int32_t (&a)[3], const int32_t (&b)[3], const int32_t (&c)[3], int64_t (&sum)[3]) {
  const int32_t ax = wsub(a[0], c[0]), ay = wsub(a[1], c[1]), az = wsub(a[2], c[2]);
  const int32_t bx = wsub(b[0], c[0]), by = wsub(b[1], c[1]), bz = wsub(b[2], c[2]);
  sum[0] = wadd64(sum[0], (int64_t)wsub(mul_w<M24>(ay, bz), mul_w<M24>(az, by)));
  sum[1] = wadd64(sum[1], (int64_t)wsub(mul_w<M24>(az, bx), mul_w<M24>(ax, bz)));
  sum[2] = wadd64(sum[2], (int64_t)wsub(mul_w<M24>(ax, by), mul_w<M24>(ay, bx)));
}

// Fan sum of cross(pos[next] - pos_c, pos[prev] - pos_c) over every face around the vertex of corner c
// (mesh_normal_prediction.rs:22-44: i32 wrapping terms, i64 wrapping sum — order-independent).  The reference swings
// left to the fan start and then right; the same set of faces is reached by swinging right AND left from c at the
// same time until the walkers meet (closed fan) or both fall off a boundary / seam (opp = NONE).  Swinging right from a
// face (next = A, prev = B) lands in the face (next = new vertex, prev = A); swinging left lands in (next = B, prev =
// new vertex): one new vertex per face.  The three dependent fetches of a face (opp hop → rank of the new vertex → its
// position) are software-pipelined across rounds, so a fan of valence v costs ≈ v/2 + 3 memory latencies, not 3v.
template <bool PACKED>
__device__ __forceinline__ void fan_normal_sum(uint32_t c, const uint32_t* __restrict__ opp, const uint32_t* __restrict__ c2r_pos,
                                               const void* __restrict__ qs_pos, const int32_t (&Pc)[3], const int32_t (&Pn)[3], const int32_t (&Pp)[3],
                                               int64_t (&sum)[3], uint32_t face_stride = 3u) {
  constexpr int fmt = PACKED ? QF_P64 : QF_I32;
  add_face_normal<PACKED>(Pn, Pp, Pc, sum);
  uint32_t curR = c, curL = c;
  bool actR = true, actL = true;
  uint32_t s1R = kNoneD, s1L = kNoneD;          // stage 1: corner of the new vertex, rank not fetched yet
  uint32_t s2R = 0, s2L = 0; bool v2R = false, v2L = false;   // stage 2: rank fetched, position not yet
  int32_t WR[3], WL[3]; bool v3R = false, v3L = false;        // stage 3: position fetched, face not added yet
  int32_t A[3] = {Pn[0], Pn[1], Pn[2]}, B[3] = {Pp[0], Pp[1], Pp[2]};
  for (uint32_t guard = 0; guard < (1u << 24); ++guard) {   // (a consistent table terminates by itself; the guard bounds a malformed one)
    if (!(actR || actL || s1R != kNoneD || s1L != kNoneD || v2R || v2L || v3R || v3L)) break;
    // ---- issue: next hops, ranks of last round's corners, positions of last round's ranks
    const uint32_t oR = actR ? opp[cidx(cprev(curR), face_stride)] : kNoneD;
    const uint32_t oL = actL ? opp[cidx(cnext(curL), face_stride)] : kNoneD;
    const uint32_t rkR = s1R != kNoneD ? c2r_pos[cidx(s1R, face_stride)] : kNoneD;
    const uint32_t rkL = s1L != kNoneD ? c2r_pos[cidx(s1L, face_stride)] : kNoneD;
    int32_t XR[3] = {0, 0, 0}, XL[3] = {0, 0, 0};
    if (v2R) load_pos_fmt(qs_pos, fmt, s2R, XR);
    if (v2L) load_pos_fmt(qs_pos, fmt, s2L, XL);
    // ---- retire stage 3 (positions fetched in the previous round)
    if (v3R) { add_face_normal<PACKED>(WR, A, Pc, sum); A[0] = WR[0]; A[1] = WR[1]; A[2] = WR[2]; }
    if (v3L) { add_face_normal<PACKED>(B, WL, Pc, sum); B[0] = WL[0]; B[1] = WL[1]; B[2] = WL[2]; }
    // ---- advance the pipeline
    v3R = v2R; v3L = v2L;
    WR[0] = XR[0]; WR[1] = XR[1]; WR[2] = XR[2]; WL[0] = XL[0]; WL[1] = XL[1]; WL[2] = XL[2];
    v2R = s1R != kNoneD; v2L = s1L != kNoneD;
    s2R = rkR; s2L = rkL;
    s1R = kNoneD; s1L = kNoneD;
    // ---- walkers: right first, then left against the updated right walker
    if (actR) {
      if (oR == kNoneD) actR = false;
      else {
        const uint32_t nR = cprev(oR);
        if (nR == curL) { actR = false; actL = false; }   // the next right face is the left walker's face: closed
        else { s1R = oR; curR = nR; }
      }
    }
    if (actL) {
      if (oL == kNoneD) actL = false;
      else {
        const uint32_t nL = cnext(oL);
        if (nL == curR) { actR = false; actL = false; }
        else { s1L = oL; curL = nL; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fan rows: the corner table of a seam-free mesh re-laid out per CODED VERTEX, in coding order (built once per job by
// k_build_fans, like c2r).  Row i = the ranks of the 1-ring of sequence entry i in fan order:
//   fan[8i+0] = a = rank of next(c), fan[8i+1] = b = rank of prev(c)                (c = seq[i])
//   then the new vertices met swinging right from c (one per face), then those met swinging left (open fans only);
//   a closed fan's last right vertex is b itself and is not stored again.
//   hdr[i] = faces_right | faces_left << 8 | closed << 16 | overflow << 17,   apex[i] = rank across the edge opposite c.
// The sweep then reads 40 contiguous bytes per entry instead of chasing `opp`/`c2r` (two dependent, two-thirds-empty
// gathers per fan face), and all its position gathers are independent.  Rows that do not fit (valence > 8) set
// `overflow` and take the corner-table walk.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kFanSlots = 8;
__device__ __forceinline__ void build_fan_row(uint32_t i, const uint32_t* __restrict__ seq, const uint32_t* __restrict__ c2r, const uint32_t* __restrict__ opp,
                                              uint32_t* __restrict__ hdr, uint32_t* __restrict__ apex, uint32_t* __restrict__ fan, int centre_in_apex) {
  {
    const uint32_t c = seq[i];
    uint32_t row[kFanSlots];
#pragma unroll
    for (uint32_t k = 0; k < kFanSlots; ++k) row[k] = kNoneD;
    row[0] = c2r[cnext(c)];
    row[1] = c2r[cprev(c)];
    const uint32_t o = opp[c];
    apex[i] = centre_in_apex ? c2r[c] : ((o != kNoneD) ? c2r[o] : kNoneD);
    uint32_t stored = 2, faces_r = 0, faces_l = 0;
    bool closed = false, overflow = false;
    uint32_t pending = kNoneD;   // the right vertex met last; written once it is known not to be the closing one (= b)
    for (uint32_t cur = c, guard = 0; guard < (1u << 24); ++guard) {
      const uint32_t o2 = opp[cprev(cur)];
      if (o2 == kNoneD) break;
      cur = cprev(o2);
      if (cur == c) { closed = true; break; }
      if (pending != kNoneD) {
        if (stored < kFanSlots) {
#pragma unroll
          for (uint32_t k = 2; k < kFanSlots; ++k) if (k == stored) row[k] = pending;
        } else overflow = true;
        ++stored;
      }
      pending = c2r[o2];
      ++faces_r;
    }
    if (!closed && pending != kNoneD) {   // open fan: the last right vertex is a real ring vertex
      if (stored < kFanSlots) {
#pragma unroll
        for (uint32_t k = 2; k < kFanSlots; ++k) if (k == stored) row[k] = pending;
      } else overflow = true;
      ++stored;
    }
    if (!closed) {
      for (uint32_t cur = c, guard = 0; guard < (1u << 24); ++guard) {
        const uint32_t o2 = opp[cnext(cur)];
        if (o2 == kNoneD) break;
        cur = cnext(o2);
        if (cur == c) break;
        const uint32_t w = c2r[o2];
        if (stored < kFanSlots) {
#pragma unroll
          for (uint32_t k = 2; k < kFanSlots; ++k) if (k == stored) row[k] = w;
        } else overflow = true;
        ++stored;
        ++faces_l;
      }
    }
    if (faces_r > 255u || faces_l > 255u) overflow = true;
    hdr[i] = (faces_r & 255u) | ((faces_l & 255u) << 8) | (closed ? 1u << 16 : 0u) | (overflow ? 1u << 17 : 0u);
    uint4* dst = reinterpret_cast<uint4*>(fan + (size_t)i * kFanSlots);
    dst[0] = make_uint4(row[0], row[1], row[2], row[3]);
    dst[1] = make_uint4(row[4], row[5], row[6], row[7]);
  }
}
__global__ __launch_bounds__(kBlock) void k_build_fans(const uint32_t* __restrict__ seq, uint32_t n, const uint32_t* __restrict__ c2r,
                                                       const uint32_t* __restrict__ opp, uint32_t* __restrict__ hdr, uint32_t* __restrict__ apex,
                                                       uint32_t* __restrict__ fan, int centre_in_apex) {
  const uint32_t blk_ = blockIdx.x, nblk_ = gridDim.x;
  DMI_FOR_SEQUENCE(i, n) build_fan_row(i, seq, c2r, opp, hdr, apex, fan, centre_in_apex);
}
// the fan rows of many (small) tables in one launch: entry g of the concatenation belongs to the last item whose offset is ≤ g
__global__ __launch_bounds__(kBlock) void k_build_fans_batch(const FanItem* __restrict__ items, uint32_t n_items, uint32_t total) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < total; g += gridDim.x * kBlock) {
    uint32_t lo = 0, hi = n_items;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (items[mid].off <= g) lo = mid; else hi = mid; }
    const FanItem it = items[lo];
    build_fan_row(g - it.off, it.seq, it.c2r, it.opp, it.hdr, it.apex, it.fan, (int)it.centre_in_apex);
  }
}

// Fan rows from FACE RECORDS (round 6: one-shot jobs in the mesh's own face order).  A swing lands in a face whose record holds both things the walk wants
// from it — the rank of the vertex it just reached and the opposite corner it leaves by: one 32-byte read per swing, where build_fan_row reads `opp` and
// `c2r` on two different lines (51.7 M L1 → L2 requests and 346 µs per 10M faces).  Rows and headers are build_fan_row's, entry for entry.
__device__ __forceinline__ uint32_t pick3(const uint4& v, uint32_t k) { return k == 0u ? v.x : (k == 1u ? v.y : v.z); }
__device__ __forceinline__ void build_fan_row_rec(uint32_t i, const uint32_t* __restrict__ seq, const uint4* __restrict__ frec, uint32_t* __restrict__ hdr, uint32_t* __restrict__ apex,
                                                  uint32_t* __restrict__ fan) {
  const uint32_t c = seq[i];
  const uint32_t f0 = c / 3u, k0 = c - 3u * f0;
  const uint4 r0 = frec[(size_t)f0 * 2], o0 = frec[(size_t)f0 * 2 + 1];
  uint32_t row[kFanSlots];
#pragma unroll
  for (uint32_t k = 0; k < kFanSlots; ++k) row[k] = kNoneD;
  const uint32_t kn = k0 == 2u ? 0u : k0 + 1u, kp = k0 == 0u ? 2u : k0 - 1u;
  row[0] = pick3(r0, kn);
  row[1] = pick3(r0, kp);
  const uint32_t o = pick3(o0, k0);
  uint32_t apex_rank = kNoneD;
  if (o != kNoneD) { const uint32_t fo = o / 3u; apex_rank = pick3(frec[(size_t)fo * 2], o - 3u * fo); }   // (off the walk's dependency chain)
  uint32_t stored = 2, faces_r = 0, faces_l = 0;
  bool closed = false, overflow = false;
  uint32_t pending = kNoneD;
  uint32_t o2 = pick3(o0, kp);                       // opp[cprev(c)]: the right swing leaves the face by the edge opposite the previous corner
  for (uint32_t guard = 0; guard < (1u << 24) && o2 != kNoneD; ++guard) {
    const uint32_t f2 = o2 / 3u, k2 = o2 - 3u * f2;
    const uint32_t kc = k2 == 0u ? 2u : k2 - 1u;     // cur = cprev(o2)
    if (f2 == f0 && kc == k0) { closed = true; break; }
    const uint4 r2 = frec[(size_t)f2 * 2], q2 = frec[(size_t)f2 * 2 + 1];
    if (pending != kNoneD) {
      if (stored < kFanSlots) {
#pragma unroll
        for (uint32_t k = 2; k < kFanSlots; ++k) if (k == stored) row[k] = pending;
      } else overflow = true;
      ++stored;
    }
    pending = pick3(r2, k2);
    ++faces_r;
    o2 = pick3(q2, kc == 0u ? 2u : kc - 1u);         // opp[cprev(cur)]
  }
  if (!closed && pending != kNoneD) {
    if (stored < kFanSlots) {
#pragma unroll
      for (uint32_t k = 2; k < kFanSlots; ++k) if (k == stored) row[k] = pending;
    } else overflow = true;
    ++stored;
  }
  if (!closed) {
    o2 = pick3(o0, kn);                              // opp[cnext(c)]
    for (uint32_t guard = 0; guard < (1u << 24) && o2 != kNoneD; ++guard) {
      const uint32_t f2 = o2 / 3u, k2 = o2 - 3u * f2;
      const uint32_t kc = k2 == 2u ? 0u : k2 + 1u;   // cur = cnext(o2)
      if (f2 == f0 && kc == k0) break;
      const uint4 r2 = frec[(size_t)f2 * 2], q2 = frec[(size_t)f2 * 2 + 1];
      const uint32_t w = pick3(r2, k2);
      if (stored < kFanSlots) {
#pragma unroll
        for (uint32_t k = 2; k < kFanSlots; ++k) if (k == stored) row[k] = w;
      } else overflow = true;
      ++stored;
      ++faces_l;
      o2 = pick3(q2, kc == 2u ? 0u : kc + 1u);       // opp[cnext(cur)]
    }
  }
  if (faces_r > 255u || faces_l > 255u) overflow = true;
  apex[i] = apex_rank;
  hdr[i] = (faces_r & 255u) | ((faces_l & 255u) << 8) | (closed ? 1u << 16 : 0u) | (overflow ? 1u << 17 : 0u);
  uint4* dst = reinterpret_cast<uint4*>(fan + (size_t)i * kFanSlots);
  dst[0] = make_uint4(row[0], row[1], row[2], row[3]);
  dst[1] = make_uint4(row[4], row[5], row[6], row[7]);
}
__global__ __launch_bounds__(kBlock) void k_build_fans_rec(const uint32_t* __restrict__ seq, uint32_t n, const uint4* __restrict__ frec, uint32_t* __restrict__ hdr,
                                                           uint32_t* __restrict__ apex, uint32_t* __restrict__ fan) {
  const uint32_t blk_ = blockIdx.x, nblk_ = gridDim.x;
  DMI_FOR_SEQUENCE(i, n) build_fan_row_rec(i, seq, frec, hdr, apex, fan);
}

// the stores of one texture-coordinate entry (orientation flag + wrapped-difference symbols)
__device__ __forceinline__ void uv_emit(const FusedArgs& a, uint32_t i, const int32_t (&cu)[2], int32_t pred0, int32_t pred1, uint8_t oflag, const WrapParams& wu, bool s16_uv) {
  __builtin_nontemporal_store(oflag, &a.orient[i]);
  const uint32_t s0 = wrap_symbol(cu[0], pred0, wu), s1 = wrap_symbol(cu[1], pred1, wu);
  if (s16_uv) DMI_SYM_STORE(s0 | (s1 << 16), static_cast<uint32_t*>(a.sym_uv) + i);   // (symbols < 2^16: one 4-byte store)
  else { store_sym(a.sym_uv, false, (size_t)i * 2, s0); store_sym(a.sym_uv, false, (size_t)i * 2 + 1, s1); }
}
// The texture-coordinate entries a fused sweep deferred (operands outside its f64 tier): the general i64 form, entry by entry.  Every
// listed entry has both neighbours coded (the sweep only defers inside `both`); positions and UVs are the sweep's (packed or plain).
__device__ __forceinline__ void k_texcoord_fixup_body(const FusedArgs& a, const uint32_t blk_, const uint32_t nblk_) {
  const uint32_t count = min(a.fix_count[0], a.n);
  if (count == 0) return;
  const bool packed = a.packed != 0u, s16_uv = (a.sym16 & 4u) != 0u;
  const WrapParams wu = wrap_params(a.mm_uv);
  auto load_uv = [&](uint32_t r, int32_t (&out)[2]) {
    if (packed) { const uint32_t v = static_cast<const uint32_t*>(a.qs_uv)[r]; out[0] = (int32_t)(v & 0xFFFFu); out[1] = (int32_t)(v >> 16); }
    else { const int32_t* q = static_cast<const int32_t*>(a.qs_uv) + (size_t)r * 2; out[0] = q[0]; out[1] = q[1]; }
  };
  for (uint32_t k = blk_ * kBlock + threadIdx.x; k < count; k += nblk_ * kBlock) {
    const uint32_t i = a.fix_list[k];
    uint32_t rn, rp;
    if (a.fan_hdr[i] & (1u << 17)) { const uint32_t c = a.seq[i]; rn = a.c2r[cidx(cnext(c), a.face_stride)]; rp = a.c2r[cidx(cprev(c), a.face_stride)]; }
    else { rn = a.fan[(size_t)i * kFanSlots]; rp = a.fan[(size_t)i * kFanSlots + 1]; }
    int32_t Pc[3], Pn[3], Pp[3], cu[2], nu[2], pu[2];
    const int fmt = packed ? QF_P64 : QF_I32;
    load_pos_fmt(a.qs_pos, fmt, i, Pc); load_pos_fmt(a.qs_pos, fmt, rn, Pn); load_pos_fmt(a.qs_pos, fmt, rp, Pp);
    load_uv(i, cu); load_uv(rn, nu); load_uv(rp, pu);
    int32_t pred0 = 0, pred1 = 0;
    uint8_t oflag = 0;
    if (!texcoord_predict_general_impl(cu, nu, pu, Pc, Pn, Pp, pred0, pred1, oflag)) { oflag = 0; pred0 = nu[0]; pred1 = nu[1]; }   // (rn < i: the sweep's fallback)
    uv_emit(a, i, cu, pred0, pred1, oflag, wu, s16_uv);
  }
}

// HAS_POS = false: the normal attribute alone, on its own (seam) table — `c2r` is then the POSITION table's corner →
// rank array (fan rows hold position ranks), `opp` the normal table's, and apex[i] is the rank of the fan's centre.
// PACKED: the positions are QF_P64 (and, with HAS_POS, the normals QF_B16 and the texture coordinates QF_H32 — the three
// attributes of a fused sweep are packed together or not at all); a lone normal attribute's own values stay QF_I32.
template <bool HAS_POS, bool HAS_NRM, bool HAS_UV, bool PACKED, bool STAGED = false>
__device__ __forceinline__ void k_predict_fused_body(const FusedArgs& a, const uint32_t blk_, const uint32_t nblk_) {
  const uint32_t* __restrict__ seq = a.seq;
  const uint32_t* __restrict__ c2r = a.c2r;
  const uint32_t* __restrict__ opp = a.opp;
  const void* __restrict__ qs_pos = a.qs_pos;
  constexpr int pos_fmt = PACKED ? QF_P64 : QF_I32;
  constexpr bool own_packed = PACKED && HAS_POS;
  const uint32_t n = a.n;
  const bool s16_pos = (a.sym16 & 1u) != 0u, s16_nrm = (a.sym16 & 2u) != 0u, s16_uv = (a.sym16 & 4u) != 0u;
  WrapParams wp{}, wu{};
  if (HAS_POS) wp = wrap_params(a.mm_pos);
  if (HAS_UV) wu = wrap_params(a.mm_uv);
  uint32_t n_false = 0;
  auto load_uv = [&](uint32_t r, int32_t (&out)[2]) {
    if (own_packed) { const uint32_t v = static_cast<const uint32_t*>(a.qs_uv)[r]; out[0] = (int32_t)(v & 0xFFFFu); out[1] = (int32_t)(v >> 16); }
    else { const int32_t* q = static_cast<const int32_t*>(a.qs_uv) + (size_t)r * 2; out[0] = q[0]; out[1] = q[1]; }
  };
  // The sweep of one entry is two dependent memory round trips (its fan row, then the positions the row names) and then arithmetic.  With
  // DMI_SWEEP_PREFETCH the first round trip of a block's NEXT chunk is issued before the current chunk's arithmetic (12 registers).
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifdef DMI_SWEEP_PREFETCH
  constexpr bool kPrefetch = PACKED && HAS_POS;
#else
  constexpr bool kPrefetch = false;
#endif
  // STAGED (round 3, the single-mesh launches): the same prefetch without registers — the next chunk's level-1 data (fan row, header, apex
  // rank, own packed position: 48 B per entry) goes global → LDS by `global_load_lds` (no VGPR destination) while the current chunk computes;
  // every lane stages and later reads its OWN entry, so the only ordering needed is the issuing wave's vmcnt (no barrier).  3 KB per
  // wavefront.  With the kernel held at 8 waves per SIMD (amdgpu_waves_per_eu: 64 VGPRs, 28 SGPRs parked in VGPR lanes) the 10M-triangle
  // sweep takes 119.3 µs instead of 130.2 (pn 85.1 → 80.1, pu 85.7 → 81.9); at the 7 waves the compiler picks by itself only −3 %.  The
  // batch launches (`_multi`: one or two chunks per block, nothing to run ahead of) lose 23 % with it and keep the direct loads.
  constexpr bool kGlds = STAGED && PACKED && HAS_POS;
  constexpr bool kStaged = kPrefetch || kGlds;
  constexpr uint32_t kImg = 3072;
  __shared__ __attribute__((aligned(16))) uint8_t glds_img[kGlds ? (kBlock / 64) * kImg : 16];
  uint8_t* const wimg = glds_img + (kGlds ? (threadIdx.x >> 6) * kImg : 0u);
  const uint32_t lane_ = threadIdx.x & 63u;
  struct Level1 { uint32_t h, ro; u32x4 r0, r1; uint64_t pc; };
  auto stage_level1 = [&](uint32_t i) {
    using G = const __attribute__((address_space(1))) void*;
    using S = __attribute__((address_space(3))) void*;
    const uint32_t ii = i < n ? i : n - 1u;
    const uint8_t* row = reinterpret_cast<const uint8_t*>(a.fan + (size_t)ii * kFanSlots);
    const uint32_t* pc32 = reinterpret_cast<const uint32_t*>(static_cast<const uint64_t*>(qs_pos) + ii);
    // (aux 2 = nt, as the direct form's nontemporal loads: the once-read rows must not push the raw values of the next step out of the MALL —
    //  with the default policy the NEXT launch of k_value_ranges took 32.1 µs instead of 29.3)
    __builtin_amdgcn_global_load_lds((G)row, (S)wimg, 16, 0, 2);
    __builtin_amdgcn_global_load_lds((G)(row + 16), (S)(wimg + 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((G)&a.fan_hdr[ii], (S)(wimg + 2048), 4, 0, 2);
    __builtin_amdgcn_global_load_lds((G)&a.fan_apex[ii], (S)(wimg + 2304), 4, 0, 2);
    __builtin_amdgcn_global_load_lds((G)pc32, (S)(wimg + 2560), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((G)(pc32 + 1), (S)(wimg + 2816), 4, 0, 0);
  };
  auto take_level1 = [&](Level1& o) {
    // the wave's own DMAs have landed (nothing else orders a ds_read behind them; a counted vmcnt(5) that leaves the previous chunk's
    // stores in flight measured the same)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    o.r0 = *reinterpret_cast<const u32x4*>(wimg + lane_ * 16u);
    o.r1 = *reinterpret_cast<const u32x4*>(wimg + 1024u + lane_ * 16u);
    o.h = *reinterpret_cast<const uint32_t*>(wimg + 2048u + lane_ * 4u);
    o.ro = *reinterpret_cast<const uint32_t*>(wimg + 2304u + lane_ * 4u);
    const uint32_t lo = *reinterpret_cast<const uint32_t*>(wimg + 2560u + lane_ * 4u), hi = *reinterpret_cast<const uint32_t*>(wimg + 2816u + lane_ * 4u);
    o.pc = (uint64_t)lo | ((uint64_t)hi << 32);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // … and have been read before the next DMA overwrites the image
  };
  auto fetch_level1 = [&](uint32_t i, Level1& o) {
    o.h = __builtin_nontemporal_load(&a.fan_hdr[i]);
    o.ro = __builtin_nontemporal_load(&a.fan_apex[i]);
    const u32x4* row4 = reinterpret_cast<const u32x4*>(a.fan + (size_t)i * kFanSlots);
    o.r0 = __builtin_nontemporal_load(&row4[0]); o.r1 = __builtin_nontemporal_load(&row4[1]);
    o.pc = static_cast<const uint64_t*>(qs_pos)[i];
  };
  const uint32_t nch_ = (n + kBlock - 1) / kBlock, per_ = (nch_ + 7u) / 8u, xcd_ = blk_ & 7u, end_ = min(nch_, (xcd_ + 1u) * per_), stride_ = nblk_ >> 3;
  uint32_t ch_ = xcd_ * per_ + (blk_ >> 3);   // (DMI_FOR_SEQUENCE's chunk walk, spelled out)
  Level1 cur{}, nxt{};
  if (kPrefetch && ch_ < end_ && ch_ * kBlock + threadIdx.x < n) fetch_level1(ch_ * kBlock + threadIdx.x, cur);
  if (kGlds && ch_ < end_) stage_level1(ch_ * kBlock + threadIdx.x);
  for (; ch_ < end_; ch_ += stride_) {
    const uint32_t i = ch_ * kBlock + threadIdx.x;
    if (kGlds) { take_level1(cur); if (ch_ + stride_ < end_) stage_level1((ch_ + stride_) * kBlock + threadIdx.x); }
    if (kPrefetch) { const uint32_t i2 = (ch_ + stride_) * kBlock + threadIdx.x; if (ch_ + stride_ < end_ && i2 < n) fetch_level1(i2, nxt); }
    [[maybe_unused]] bool deferred = false;
    if (i < n) {
    uint32_t rn, rp, ro;
    int32_t Pc[3], Pn[3] = {0, 0, 0}, Pp[3] = {0, 0, 0}, Po[3] = {0, 0, 0}, Plast[3] = {0, 0, 0};
    int64_t sum[3] = {0, 0, 0};
    uint32_t h;
    if (kStaged) { h = cur.h; ro = cur.ro; unpack_p64(cur.pc, Pc); }
    else {
      h = __builtin_nontemporal_load(&a.fan_hdr[i]);
      ro = __builtin_nontemporal_load(&a.fan_apex[i]);   // HAS_POS: rank across the edge opposite c; else: rank of the centre
      load_pos_fmt(qs_pos, pos_fmt, HAS_POS ? i : ro, Pc);
    }
    if (!(h & (1u << 17))) {
      // ---- fan row: every rank of the 1-ring in one 32-byte read, every position gather independent ----
      u32x4 r0, r1;
      if (kStaged) { r0 = cur.r0; r1 = cur.r1; }
      else { const u32x4* row4 = reinterpret_cast<const u32x4*>(a.fan + (size_t)i * kFanSlots); r0 = __builtin_nontemporal_load(&row4[0]); r1 = __builtin_nontemporal_load(&row4[1]); }
      const uint32_t row[kFanSlots] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      rn = row[0]; rp = row[1];
      const uint32_t faces_r = h & 255u, faces_l = (h >> 8) & 255u;
      const bool closed = (h >> 16) & 1u;
      const uint32_t stored_r = (closed && faces_r) ? faces_r - 1u : faces_r;
      const uint32_t cnt = HAS_NRM ? 2u + stored_r + faces_l : 2u;
      // every gather of the ring is issued before any is used; packed positions stay packed (one register pair each) until the
      // face they belong to is formed — the sweep's speed follows its wave count (7 per SIMD at 69 VGPRs, 20 % slower at 5)
      [[maybe_unused]] uint64_t Pq[kFanSlots];
      [[maybe_unused]] int32_t P[kFanSlots][3];
      const bool need_np = HAS_NRM || (HAS_POS && rn < i && rp < i);
#pragma unroll
      for (uint32_t k = 0; k < kFanSlots; ++k) {
        const bool want = k < cnt && (k >= 2 || need_np);
        if (PACKED) Pq[k] = (want && row[k] != kNoneD) ? static_cast<const uint64_t*>(qs_pos)[row[k]] : 0ull;
        else if (want) load_pos_fmt(qs_pos, pos_fmt, row[k], P[k]);
        else { P[k][0] = 0; P[k][1] = 0; P[k][2] = 0; }
      }
      auto ring = [&](uint32_t k, int32_t (&o)[3]) { if (PACKED) unpack_p64(Pq[k], o); else { o[0] = P[k][0]; o[1] = P[k][1]; o[2] = P[k][2]; } };
      ring(0, Pn); ring(1, Pp);
      if (HAS_NRM) {
        // faces: (next, prev) = (a, b); right of it (w1, a), (w2, w1), …, closing (b, w_last); left of it (b, u1), (u1, u2), …
        // (edge vectors ring − centre are formed once per ring vertex and serve the two faces it borders; R / L = the edge the right /
        //  left walk stands on — the closing face of a closed fan runs from b, which no left face has replaced, to the last right vertex)
        int32_t R[3] = {wsub(Pn[0], Pc[0]), wsub(Pn[1], Pc[1]), wsub(Pn[2], Pc[2])}, L[3] = {wsub(Pp[0], Pc[0]), wsub(Pp[1], Pc[1]), wsub(Pp[2], Pc[2])};
        add_edge_cross<PACKED>(R, L, sum);
#pragma unroll
        for (uint32_t k = 2; k < kFanSlots; ++k) {
          if (k < cnt) {
            int32_t W[3];
            ring(k, W);
            W[0] = wsub(W[0], Pc[0]); W[1] = wsub(W[1], Pc[1]); W[2] = wsub(W[2], Pc[2]);
            if (k < 2u + stored_r) { add_edge_cross<PACKED>(W, R, sum); R[0] = W[0]; R[1] = W[1]; R[2] = W[2]; }
            else { add_edge_cross<PACKED>(L, W, sum); L[0] = W[0]; L[1] = W[1]; L[2] = W[2]; }
          }
        }
        if (closed && faces_r) add_edge_cross<PACKED>(L, R, sum);
      }
    } else {
      // ---- row overflow (valence > 8): walk the corner table ----
      const uint32_t c = seq[i], nc = cnext(c), pc = cprev(c);
      rn = c2r[cidx(nc, a.face_stride)]; rp = c2r[cidx(pc, a.face_stride)];
      if (HAS_NRM || (HAS_POS && rn < i && rp < i)) { load_pos_fmt(qs_pos, pos_fmt, rn, Pn); load_pos_fmt(qs_pos, pos_fmt, rp, Pp); }
      if (HAS_NRM) fan_normal_sum<PACKED>(c, opp, c2r, qs_pos, Pc, Pn, Pp, sum, a.face_stride);
    }
    const bool both = HAS_POS && rn < i && rp < i;
    if (HAS_POS) {
      // ---- positions: mesh_parallelogram_prediction.rs:186-237 + wrapped difference ----
      const bool have = both && ro < i;   // (no opposite corner ⇒ ro == NONE ⇒ false)
      if (have) load_pos_fmt(qs_pos, pos_fmt, ro, Po);
      else if (i > 0) load_pos_fmt(qs_pos, pos_fmt, i - 1u, Plast);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int32_t pred = have ? wsub(wadd(Pn[k], Pp[k]), Po[k]) : Plast[k];   // Q15: previously coded vertex, 0 for the first entry
        store_sym(a.sym_pos, s16_pos, (size_t)i * 3 + k, wrap_symbol(Pc[k], pred, wp));
      }
    }
    // ---- normals: mesh_normal_prediction.rs:22-44,75-144 + oct_orthogonal.rs (before the texture coordinates: the fan sums die here) ----
    if (HAS_NRM) {
      // Fan sum of cross(pos[next] - pos_c, pos[prev] - pos_c).  Swinging right from a face (next = A, prev = B) lands in
      // the face (next = new vertex, prev = A); swinging left lands in (next = B, prev = new vertex).
      int64_t sum0 = sum[0], sum1 = sum[1], sum2 = sum[2];
      const int64_t upper = 1ll << 29;
      const int64_t abs_sum = wadd64(wadd64(wabs64(sum0), wabs64(sum1)), wabs64(sum2));
      if (abs_sum > upper) {
        const int64_t quot = abs_sum / upper;
        sum0 = wdiv64(sum0, quot); sum1 = wdiv64(sum1, quot); sum2 = wdiv64(sum2, quot);
      }
      const int32_t n0 = (int32_t)sum0, n1 = (int32_t)sum1, n2 = (int32_t)sum2;
      int32_t p0 = 0, p1 = 0;
      if (!(n0 == 0 && n1 == 0 && n2 == 0)) oct_quantize((float)n0, (float)n1, (float)n2, p0, p1);
      int32_t a0, a1;
      if (own_packed) { const uint32_t v = static_cast<const uint16_t*>(a.qs_nrm)[i]; a0 = (int32_t)(v & 0xFFu); a1 = (int32_t)(v >> 8); }
      else { const int32_t* q = static_cast<const int32_t*>(a.qs_nrm) + (size_t)i * 2; a0 = q[0]; a1 = q[1]; }
      const int32_t m0 = (int32_t)(0u - (uint32_t)p0), m1 = (int32_t)(0u - (uint32_t)p1);
      const int32_t d10 = wsub(p0, a0), d11 = wsub(p1, a1), d20 = wsub(m0, a0), d21 = wsub(m1, a1);
      const int32_t dot1 = wadd(__mul24(d10, d10), __mul24(d11, d11)), dot2 = wadd(__mul24(d20, d20), __mul24(d21, d21));   // (|d| ≤ 510: exact)
      const bool flip = dot1 > dot2;   // Q8: flip negates the octahedral coordinates
      if (flip) { p0 = m0; p1 = m1; } else ++n_false;
      __builtin_nontemporal_store((uint8_t)(flip ? 1 : 0), &a.flips[i]);
      uint32_t s0, s1;
      oct_orthogonal(a0, a1, p0, p1, s0, s1);
      if (s16_nrm) DMI_SYM_STORE(s0 | (s1 << 16), static_cast<uint32_t*>(a.sym_nrm) + i);
      else { store_sym(a.sym_nrm, false, (size_t)i * 2, s0); store_sym(a.sym_nrm, false, (size_t)i * 2 + 1, s1); }
    }
    // ---- texture coordinates: mesh_prediction_for_texture_coordinates.rs:51-81,107-219 ----
    if (HAS_UV) {
      int32_t cu[2];
      load_uv(i, cu);
      int32_t pred0 = 0, pred1 = 0;
      uint8_t oflag = 0;
      bool done = false;
      int32_t nu[2] = {0, 0};
      if (rn < i) load_uv(rn, nu);
      if (both) {
        int32_t pu[2];
        load_uv(rp, pu);
        const int st = texcoord_try(cu, nu, pu, Pc, Pn, Pp, pred0, pred1, oflag);
        done = st == 1;
        deferred = st == 2;   // operands outside the f64 tier: k_texcoord_fixup predicts this entry with the general form
      }
      if (!done) {
        oflag = 0;
        if (rn < i) { pred0 = nu[0]; pred1 = nu[1]; }
        else if (i > 0) { int32_t lu[2]; load_uv(i - 1u, lu); pred0 = lu[0]; pred1 = lu[1]; }
        else { pred0 = 0; pred1 = 0; }
      }
      if (!deferred) uv_emit(a, i, cu, pred0, pred1, oflag, wu, s16_uv);
    }
    }
    if (HAS_UV) {   // deferred entries of this wavefront: one atomic for all of them
      const uint64_t m = __ballot(deferred);
      if (m) {
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // deferred lanes below this one
        uint32_t base = 0;
        if (lane == 0 && deferred) base = atomicAdd(a.fix_count, (uint32_t)__popcll(m));
        base = __shfl(base, (int)__builtin_ctzll(m), 64);
        if (deferred) a.fix_list[base + lane] = i;
      }
    }
    if (kPrefetch) cur = nxt;
  }
  if (HAS_NRM) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_false += __shfl_down(n_false, off, 64);
    __shared__ uint32_t wave_false[kBlock / 64];
    if ((threadIdx.x & 63) == 0) wave_false[threadIdx.x >> 6] = n_false;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) t += wave_false[w];
            if (a.flip_partials) a.flip_partials[blk_] = t;   // (summed by the histogram launch: no same-address atomic per block)
      else if (t) atomicAdd(&a.counters[0], t);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same sweep with its prediction neighbourhoods staged through LDS (packed values only: positions QF_P64, normals QF_B16,
// texture coordinates QF_H32).  The 1-ring of the 256 consecutive sequence entries of a chunk lives in three short runs of the
// coding order: right around the chunk (the strip the traversal is laying down), one ring behind it and one ring ahead of it
// (Edgebreaker order: a ring is a few thousand entries).  The block reads its fan rows, finds where the two far runs start (a block
// minimum over the ranks that fall outside the near window), copies the three windows of packed positions — and the near / behind
// windows of packed texture coordinates — into LDS with coalesced loads, and every gather of the sweep then reads LDS; a rank outside
// the windows (irregular meshes, the seams of the spiral) falls back to its global gather.  Results are the fused sweep's.
// MEASURED (10M-triangle workload, MI355X): 217 µs against 155 µs for the plain packed sweep — the staging costs two barriers and a
// block reduction per 256 entries and, above all, registers (124 VGPRs against 69: 4 waves per SIMD instead of 7; capped to 80 it
// spills 3 KB per thread and takes 13 ms), while the plain sweep's gathers already land on lines its own L2 holds.  Kept behind
// DMI_FUSED_WINDOWS=1 as the recorded experiment; not the default.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kWinBack = 64, kWinFwd = 64, kWinNear = kBlock + kWinBack + kWinFwd, kWinFar = 512;
template <bool HAS_NRM, bool HAS_UV>
__device__ __forceinline__ void k_predict_window_body(const FusedArgs& a, const uint32_t blk_, const uint32_t nblk_) {
  __shared__ uint64_t wpos[kWinNear + 2 * kWinFar];
  __shared__ uint32_t wuv[HAS_UV ? kWinNear + kWinFar : 1];
  __shared__ uint32_t wmin[2][kBlock / 64];
  __shared__ uint32_t wave_false[kBlock / 64];
  const uint32_t* __restrict__ seq = a.seq;
  const uint32_t* __restrict__ c2r = a.c2r;
  const uint32_t* __restrict__ opp = a.opp;
  const uint64_t* __restrict__ qpos = static_cast<const uint64_t*>(a.qs_pos);
  const uint32_t* __restrict__ quv = static_cast<const uint32_t*>(a.qs_uv);
  const uint32_t n = a.n;
  const bool s16_pos = (a.sym16 & 1u) != 0u, s16_nrm = (a.sym16 & 2u) != 0u, s16_uv = (a.sym16 & 4u) != 0u;
  const WrapParams wp = wrap_params(a.mm_pos);
  WrapParams wu{};
  if (HAS_UV) wu = wrap_params(a.mm_uv);
  uint32_t n_false = 0;
  const uint32_t tid = threadIdx.x;
  const uint32_t nch = (n + kBlock - 1) / kBlock, per = (nch + 7u) / 8u, xcd = blk_ & 7u, end = min(nch, (xcd + 1u) * per);
  for (uint32_t ch = xcd * per + (blk_ >> 3); ch < end; ch += (nblk_ >> 3)) {
    const uint32_t base = ch * kBlock, i = base + tid;
    const bool live = i < n;
    // ---- the chunk's fan rows ----
    uint32_t h = 0, ro = kNoneD, row[kFanSlots];
#pragma unroll
    for (uint32_t k = 0; k < kFanSlots; ++k) row[k] = kNoneD;
    if (live) {
      ro = a.fan_apex[i];
      const uint4* row4 = reinterpret_cast<const uint4*>(a.fan + (size_t)i * kFanSlots);
      const uint4 r0 = row4[0], r1 = row4[1];
      row[0] = r0.x; row[1] = r0.y; row[2] = r0.z; row[3] = r0.w; row[4] = r1.x; row[5] = r1.y; row[6] = r1.z; row[7] = r1.w;
    }
    // ---- where the far windows start: block minimum of the ranks behind / ahead of the near window ----
    const uint32_t near_lo = base >= kWinBack ? base - kWinBack : 0u, near_n = min(n, base + kBlock + kWinFwd) - near_lo;
    uint32_t mb = kNoneD, mf = kNoneD;
    {
      const uint32_t near_hi = near_lo + near_n;
#pragma unroll
      for (uint32_t k = 0; k < kFanSlots; ++k) { const uint32_t r = row[k]; if (r < near_lo) mb = min(mb, r); else if (r >= near_hi && r != kNoneD) mf = min(mf, r); }
      if (ro < near_lo) mb = min(mb, ro); else if (ro >= near_hi && ro != kNoneD) mf = min(mf, ro);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { mb = min(mb, (uint32_t)__shfl_xor((int)mb, off, 64)); mf = min(mf, (uint32_t)__shfl_xor((int)mf, off, 64)); }
      if ((tid & 63u) == 0u) { wmin[0][tid >> 6] = mb; wmin[1][tid >> 6] = mf; }
      __syncthreads();
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) { mb = min(mb, wmin[0][w]); mf = min(mf, wmin[1][w]); }
    }
    const uint32_t back_lo = mb, back_n = mb == kNoneD ? 0u : min(kWinFar, near_lo - mb);   // [mb, mb + back_n) stays below the near window
    const uint32_t fwd_lo = mf, fwd_n = mf == kNoneD ? 0u : min(kWinFar, n - mf);
    // ---- stage the windows (coalesced) ----
    for (uint32_t k = tid; k < near_n; k += kBlock) wpos[k] = qpos[near_lo + k];
    for (uint32_t k = tid; k < back_n; k += kBlock) wpos[kWinNear + k] = qpos[back_lo + k];
    for (uint32_t k = tid; k < fwd_n; k += kBlock) wpos[kWinNear + kWinFar + k] = qpos[fwd_lo + k];
    if (HAS_UV) {
      for (uint32_t k = tid; k < near_n; k += kBlock) wuv[k] = quv[near_lo + k];
      for (uint32_t k = tid; k < back_n; k += kBlock) wuv[kWinNear + k] = quv[back_lo + k];
    }
    __syncthreads();
    if (live) {   // (the rows are read again rather than kept in registers across the staging: the sweep is register-bound)
      h = a.fan_hdr[i];
      ro = a.fan_apex[i];
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4* row4 = reinterpret_cast<const u32x4*>(a.fan + (size_t)i * kFanSlots);
      const u32x4 r0 = __builtin_nontemporal_load(&row4[0]), r1 = __builtin_nontemporal_load(&row4[1]);
      row[0] = r0.x; row[1] = r0.y; row[2] = r0.z; row[3] = r0.w; row[4] = r1.x; row[5] = r1.y; row[6] = r1.z; row[7] = r1.w;
    }
    auto fetch_pos = [&](uint32_t r, int32_t (&out)[3]) {
      if (r == kNoneD) { out[0] = 0; out[1] = 0; out[2] = 0; return; }
      uint32_t d;
      uint64_t v;
      if ((d = r - near_lo) < near_n) v = wpos[d];
      else if ((d = r - back_lo) < back_n) v = wpos[kWinNear + d];
      else if ((d = r - fwd_lo) < fwd_n) v = wpos[kWinNear + kWinFar + d];
      else v = qpos[r];
      unpack_p64(v, out);
    };
    auto fetch_uv = [&](uint32_t r, int32_t (&out)[2]) {
      uint32_t d, v;
      if ((d = r - near_lo) < near_n) v = wuv[d];
      else if ((d = r - back_lo) < back_n) v = wuv[kWinNear + d];
      else v = quv[r];
      out[0] = (int32_t)(v & 0xFFFFu); out[1] = (int32_t)(v >> 16);
    };
    if (live) {
      uint32_t rn, rp;
      int32_t Pc[3], Pn[3] = {0, 0, 0}, Pp[3] = {0, 0, 0}, Po[3] = {0, 0, 0}, Plast[3] = {0, 0, 0};
      int64_t sum[3] = {0, 0, 0};
      fetch_pos(i, Pc);
      if (!(h & (1u << 17))) {
        rn = row[0]; rp = row[1];
        const uint32_t faces_r = h & 255u, faces_l = (h >> 8) & 255u;
        const bool closed = (h >> 16) & 1u;
        const uint32_t stored_r = (closed && faces_r) ? faces_r - 1u : faces_r;
        const uint32_t cnt = HAS_NRM ? 2u + stored_r + faces_l : 2u;
        int32_t P[kFanSlots][3];
        const bool need_np = HAS_NRM || (rn < i && rp < i);
#pragma unroll
        for (uint32_t k = 0; k < kFanSlots; ++k) {
          if (k < cnt && (k >= 2 || need_np)) fetch_pos(row[k], P[k]);
          else { P[k][0] = 0; P[k][1] = 0; P[k][2] = 0; }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) { Pn[d] = P[0][d]; Pp[d] = P[1][d]; }
        if (HAS_NRM) {
          add_face_normal<true>(Pn, Pp, Pc, sum);
          int32_t R[3] = {Pn[0], Pn[1], Pn[2]}, L[3] = {Pp[0], Pp[1], Pp[2]};
#pragma unroll
          for (uint32_t k = 2; k < kFanSlots; ++k) {
            if (k < cnt) {
              if (k < 2u + stored_r) { add_face_normal<true>(P[k], R, Pc, sum); R[0] = P[k][0]; R[1] = P[k][1]; R[2] = P[k][2]; }
              else { add_face_normal<true>(L, P[k], Pc, sum); L[0] = P[k][0]; L[1] = P[k][1]; L[2] = P[k][2]; }
            }
          }
          if (closed && faces_r) add_face_normal<true>(Pp, R, Pc, sum);
        }
      } else {
        // ---- row overflow (valence > 8): walk the corner table (global gathers) ----
        const uint32_t c = seq[i], nc = cnext(c), pc = cprev(c);
        rn = c2r[nc]; rp = c2r[pc];
        if (HAS_NRM || (rn < i && rp < i)) { fetch_pos(rn, Pn); fetch_pos(rp, Pp); }
        if (HAS_NRM) fan_normal_sum<true>(c, opp, c2r, a.qs_pos, Pc, Pn, Pp, sum);
      }
      const bool both = rn < i && rp < i;
      {
        const bool have = both && ro < i;
        if (have) fetch_pos(ro, Po);
        else if (i > 0) fetch_pos(i - 1u, Plast);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int32_t pred = have ? wsub(wadd(Pn[k], Pp[k]), Po[k]) : Plast[k];
          store_sym(a.sym_pos, s16_pos, (size_t)i * 3 + k, wrap_symbol(Pc[k], pred, wp));
        }
      }
      if (HAS_UV) {
        int32_t cu[2];
        fetch_uv(i, cu);
        int32_t pred0 = 0, pred1 = 0;
        uint8_t oflag = 0;
        bool done = false;
        int32_t nu[2] = {0, 0};
        if (rn < i) fetch_uv(rn, nu);
        if (both) {
          int32_t pu[2];
          fetch_uv(rp, pu);
          done = texcoord_predict(cu, nu, pu, Pc, Pn, Pp, pred0, pred1, oflag);
        }
        if (!done) {
          oflag = 0;
          if (rn < i) { pred0 = nu[0]; pred1 = nu[1]; }
          else if (i > 0) { int32_t lu[2]; fetch_uv(i - 1u, lu); pred0 = lu[0]; pred1 = lu[1]; }
          else { pred0 = 0; pred1 = 0; }
        }
        __builtin_nontemporal_store(oflag, &a.orient[i]);
        const uint32_t s0 = wrap_symbol(cu[0], pred0, wu), s1 = wrap_symbol(cu[1], pred1, wu);
        if (s16_uv) __builtin_nontemporal_store(s0 | (s1 << 16), static_cast<uint32_t*>(a.sym_uv) + i);
        else { store_sym(a.sym_uv, false, (size_t)i * 2, s0); store_sym(a.sym_uv, false, (size_t)i * 2 + 1, s1); }
      }
      if (HAS_NRM) {
        int64_t sum0 = sum[0], sum1 = sum[1], sum2 = sum[2];
        const int64_t upper = 1ll << 29;
        const int64_t abs_sum = wadd64(wadd64(wabs64(sum0), wabs64(sum1)), wabs64(sum2));
        if (abs_sum > upper) {
          const int64_t quot = abs_sum / upper;
          sum0 = wdiv64(sum0, quot); sum1 = wdiv64(sum1, quot); sum2 = wdiv64(sum2, quot);
        }
        const int32_t n0 = (int32_t)sum0, n1 = (int32_t)sum1, n2 = (int32_t)sum2;
        int32_t p0 = 0, p1 = 0;
        if (!(n0 == 0 && n1 == 0 && n2 == 0)) oct_quantize((float)n0, (float)n1, (float)n2, p0, p1);
        const uint32_t v = static_cast<const uint16_t*>(a.qs_nrm)[i];
        const int32_t a0 = (int32_t)(v & 0xFFu), a1 = (int32_t)(v >> 8);
        const int32_t m0 = (int32_t)(0u - (uint32_t)p0), m1 = (int32_t)(0u - (uint32_t)p1);
        const int32_t d10 = wsub(p0, a0), d11 = wsub(p1, a1), d20 = wsub(m0, a0), d21 = wsub(m1, a1);
        const int32_t dot1 = wadd(__mul24(d10, d10), __mul24(d11, d11)), dot2 = wadd(__mul24(d20, d20), __mul24(d21, d21));   // (|d| ≤ 510: exact)
        const bool flip = dot1 > dot2;
        if (flip) { p0 = m0; p1 = m1; } else ++n_false;
        __builtin_nontemporal_store((uint8_t)(flip ? 1 : 0), &a.flips[i]);
        uint32_t s0, s1;
        oct_orthogonal(a0, a1, p0, p1, s0, s1);
        if (s16_nrm) __builtin_nontemporal_store(s0 | (s1 << 16), static_cast<uint32_t*>(a.sym_nrm) + i);
        else { store_sym(a.sym_nrm, false, (size_t)i * 2, s0); store_sym(a.sym_nrm, false, (size_t)i * 2 + 1, s1); }
      }
    }
    __syncthreads();   // the next chunk restages the windows
  }
  if (HAS_NRM) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_false += __shfl_down(n_false, off, 64);
    if ((tid & 63) == 0) wave_false[tid >> 6] = n_false;
    __syncthreads();
    if (tid == 0) {
      uint32_t t = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) t += wave_false[w];
      if (a.flip_partials) a.flip_partials[blk_] = t;
      else if (t) atomicAdd(&a.counters[0], t);
    }
  }
}

// Per-block summary of the orientation flags so the host can stitch the count of bits and the number
// of forward transitions (mesh_prediction_for_texture_coordinates.rs:224-235) without a serial pass.
constexpr uint32_t kOrientChunk = 4096;
template <uint32_t THREADS>
__device__ __forceinline__ void orient_summary_impl(const OrientArgs& oa, const uint32_t blk_) {
  const uint8_t* __restrict__ orient = oa.orient;
  const uint32_t n = oa.n;
  uint32_t* __restrict__ summary = oa.summary;
  const uint32_t lo = blk_ * kOrientChunk, hi = min(n, lo + kOrientChunk);
  const uint32_t lane = threadIdx.x;
  // stage the chunk through LDS with 16-byte loads: the scan below is a 64-step dependent loop, and a global byte
  // load per step would expose one memory latency per step
  __shared__ __attribute__((aligned(16))) uint8_t staged[kOrientChunk];
  for (uint32_t w = lane; w < kOrientChunk / 16; w += THREADS) {
    const uint32_t at = lo + w * 16;
    if (at + 16 <= n) *reinterpret_cast<uint4*>(staged + w * 16) = *reinterpret_cast<const uint4*>(orient + at);
    else for (uint32_t b = 0; b < 16; ++b) staged[w * 16 + b] = (at + b < n) ? orient[at + b] : (uint8_t)0;
  }
  __syncthreads();
  if (THREADS > 64 && lane >= 64) return;   // the scan is one wavefront's (no barrier follows)
  uint32_t count = 0, trans = 0, first = 2, last = 2;
  for (uint32_t base = lo; base < hi; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t f = (i < hi) ? staged[i - lo] : 0;
    const unsigned long long valid = __ballot(f != 0);
    const unsigned long long ones = __ballot(f == 2);
    if (valid == 0) continue;
    // value of the previous valid lane (or `last` carried from the previous batch)
    const unsigned long long below = valid & ((1ull << lane) - 1ull);
    uint32_t prev;
    if (below) { const int pl = 63 - __clzll(below); prev = (uint32_t)((ones >> pl) & 1ull); }
    else prev = last;
    const uint32_t mine = (f == 2);
    const bool is_t = (f != 0) && (prev != 2) && (prev != mine);
    trans += (uint32_t)__popcll(__ballot(is_t));
    count += (uint32_t)__popcll(valid);
    if (first == 2) { const int fl = __ffsll((long long)valid) - 1; first = (uint32_t)((ones >> fl) & 1ull); }
    { const int ll = 63 - __clzll(valid); last = (uint32_t)((ones >> ll) & 1ull); }
  }
  if (lane == 0) {
    summary[blk_ * 4 + 0] = count;
    summary[blk_ * 4 + 1] = first;
    summary[blk_ * 4 + 2] = last;
    summary[blk_ * 4 + 3] = trans;
  }
}

__device__ __forceinline__ void k_orient_summary_body(const OrientArgs& oa, const uint32_t blk_, const uint32_t) { orient_summary_impl<64>(oa, blk_); }

// Symbol histograms of every attribute of a job in one launch (block → (attribute, slice)), the first 16K bins privatised in LDS
// (64 KiB of the CU's 160 KiB).  Few, fat blocks: each flushes its private copy once.
constexpr uint32_t kHistHotWords = 1024;   // 64 hot bins × 16 copies (see k_histogram_body)
constexpr uint32_t kLdsBins = 12288;   // 48 KiB: with the static LDS of the launch (orientation staging) below the 64 KiB a workgroup may always take
__device__ __forceinline__ void k_histogram_body(const HistArgs& args, const uint32_t blk_, const uint32_t nblk_) {
  extern __shared__ uint32_t lds_all[];
  uint32_t* lds = lds_all;
  // trailing blocks: the orientation-flag summaries of the job's fused sweep ride this launch (they depend on the same sweep as the
  // histograms and would otherwise be a 16 µs serial step of the pass)
  if (blk_ >= args.hist_blocks) { orient_summary_impl<kBlock>(args.orient, blk_ - args.hist_blocks); return; }
  int ai = 0;
  while (ai + 1 < args.count && blk_ >= args.a[ai + 1].first_block) ++ai;
  const HistAtt a = args.a[ai];
  const uint32_t block = blk_ - a.first_block;
  if (block == 0 && a.flip_partials) {   // the unflipped-normal count of the attribute's sweep: its per-block counts, summed once
    uint32_t t = 0;
    for (uint32_t b = threadIdx.x; b < a.n_flip_partials; b += kBlock) t += a.flip_partials[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    __shared__ uint32_t wave_sum[kBlock / 64];
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t sum = 0; for (int w = 0; w < kBlock / 64; ++w) sum += wave_sum[w]; a.flip_count[0] = sum; }
  }
  // the first kLdsBins bins are privatised in LDS; larger alphabets (≥ 14-bit wrapped differences) send their high — rare: the
  // residuals concentrate near zero — symbols straight to the global histogram.  (All-global atomics on a peaked distribution
  // serialise on a few addresses: 320 ms instead of 0.7 for the 150M position symbols of a 100M-triangle mesh.)
  const uint32_t lds_bins = min(a.bins, kLdsBins);
  // The residuals pile up on the smallest symbols, and LDS atomics of one wavefront to one address execute one after the other: the first
  // kHotBins bins are kept in kHotCopies copies (a lane adds to copy lane % kHotCopies), folded together at the end.
  constexpr uint32_t kHotBins = 64, kHotCopies = 16;
  uint32_t* hot = lds;   // kHistHotWords words in front of the private bins (launch_histograms sizes the dynamic LDS)
  lds += kHistHotWords;
  static_assert(kHotBins * kHotCopies == kHistHotWords, "hot copies");
  for (uint32_t b = threadIdx.x; b < lds_bins; b += kBlock) lds[b] = 0;
  for (uint32_t b = threadIdx.x; b < kHotBins * kHotCopies; b += kBlock) hot[b] = 0;
  __syncthreads();
  const uint32_t copy = (threadIdx.x & (kHotCopies - 1u)) * kHotBins;
  auto add = [&](uint32_t s) {
    if (s < kHotBins) atomicAdd(&hot[copy + s], 1u);
    else if (s < lds_bins) atomicAdd(&lds[s], 1u);
    else if (s < a.bins) atomicAdd(&a.hist[s], 1u);
    else atomicOr(a.overflow, 1u);
  };
  if (a.sym16) {   // eight 16-bit symbols per load (the symbol arrays are 16-byte aligned and padded)
    const uint64_t n8 = a.n / 8;
    const uint4* __restrict__ src = static_cast<const uint4*>(a.sym);
    for (uint64_t e = (uint64_t)block * kBlock + threadIdx.x; e < n8; e += (uint64_t)a.blocks * kBlock) {
      const uint4 v = src[e];
      add(v.x & 0xFFFFu); add(v.x >> 16); add(v.y & 0xFFFFu); add(v.y >> 16); add(v.z & 0xFFFFu); add(v.z >> 16); add(v.w & 0xFFFFu); add(v.w >> 16);
    }
    if (block == 0) for (uint64_t e = n8 * 8 + threadIdx.x; e < a.n; e += kBlock) add((uint32_t)static_cast<const uint16_t*>(a.sym)[e]);
  } else {
    const uint64_t n4 = a.n / 4;
    const uint4* __restrict__ src = static_cast<const uint4*>(a.sym);
    for (uint64_t e = (uint64_t)block * kBlock + threadIdx.x; e < n4; e += (uint64_t)a.blocks * kBlock) { const uint4 v = src[e]; add(v.x); add(v.y); add(v.z); add(v.w); }
    if (block == 0) for (uint64_t e = n4 * 4 + threadIdx.x; e < a.n; e += kBlock) add(static_cast<const uint32_t*>(a.sym)[e]);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < lds_bins; b += kBlock) {
    uint32_t v = lds[b];
    if (b < kHotBins) for (uint32_t c = 0; c < kHotCopies; ++c) v += hot[c * kHotBins + b];
    if (v) atomicAdd(&a.hist[b], v);
  }
  if (lds_bins < kHotBins)   // (an alphabet smaller than the hot range: its copies still hold counts for bins < a.bins)
    for (uint32_t b = lds_bins + threadIdx.x; b < min(a.bins, kHotBins); b += kBlock) { uint32_t v = 0; for (uint32_t c = 0; c < kHotCopies; ++c) v += hot[c * kHotBins + b]; if (v) atomicAdd(&a.hist[b], v); }
}

// ------------------------------------------------------------------------------------------------
// Decoder side: the stages of reading an attribute section back that are data-parallel.  Entropy decoding and the
// position / texture-coordinate predictions are sequential like their encoders (each value needs the ones decoded before it) and
// run on host cores (dmi_decode.cpp); the NORMAL predictor only needs the — already decoded — positions, so every entry of a
// normal attribute is reconstructed independently here, and dequantization is a scatter over corners.
//   k_decode_normals   mesh_normal_prediction.rs:22-44,75-144 re-run + oct_orthogonal.rs:23-74 inverted (the reference's own inverse
//                      is unimplemented!(): decode/attribute/inverse_prediction_transform/oct_orthogonal.rs:40)
//   k_dequantize       the inverse of quantization_coordinate_wise.rs:70-91 / octahedral_quantization.rs:49-64 (Draco's dequantizers:
//                      v = min + q · range / (2^bits - 1); octahedral (u, v) → unit vector), written to every point of the vertex
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void invert_diamond_encoder(int32_t& a, int32_t& b) {   // oct_orthogonal.rs:35-45 (sign(0) = 0 and all)
  const int32_t a0 = a, q = -isgn(wmul(a, b));
  a = wadd(wmul(q, b), wmul(isgn(a0), 127));
  b = wadd(wmul(q, a0), wmul(isgn(b), 127));
}
__device__ __forceinline__ void invert_diamond_involution(int32_t& s, int32_t& t) {   // the format's involution (inverts the map above off the axes)
  int32_t sign_s, sign_t;
  if (s >= 0 && t >= 0) { sign_s = 1; sign_t = 1; }
  else if (s <= 0 && t <= 0) { sign_s = -1; sign_t = -1; }
  else { sign_s = s > 0 ? 1 : -1; sign_t = t > 0 ? 1 : -1; }
  const int32_t cs = sign_s * 127, ct = sign_t * 127;
  int32_t us = t + t - ct, ut = s + s - cs;
  if (sign_s * sign_t >= 0) { us = -us; ut = -ut; }
  s = (us + cs) / 2;
  t = (ut + ct) / 2;
}
__device__ __forceinline__ void oct_orthogonal_inverse(int32_t p0, int32_t p1, int32_t c0, int32_t c1, int32_t& o0, int32_t& o1) {
  p0 = wsub(p0, 127); p1 = wsub(p1, 127);
  const bool inverted = wadd(iabs(p0), iabs(p1)) > 127;
  if (inverted) invert_diamond_encoder(p0, p1);
  int turns = 0;
  if (!(p0 == 0 && p1 == 0)) for (; turns < 4 && (p0 >= 0 || p1 > 0); ++turns) { const int32_t t = p0; p0 = (int32_t)(0u - (uint32_t)p1); p1 = t; }
  o0 = wadd(c0, p0); o1 = wadd(c1, p1);
  if (o0 > 127) o0 = wsub(o0, 255);
  if (o1 > 127) o1 = wsub(o1, 255);
  for (int k = 0; k < (4 - turns % 4) % 4; ++k) { const int32_t t = o0; o0 = (int32_t)(0u - (uint32_t)o1); o1 = t; }
  if (inverted) invert_diamond_involution(o0, o1);
  o0 = wadd(o0, 127); o1 = wadd(o1, 127);
}
__global__ __launch_bounds__(kBlock) void k_decode_normals(DecodeNormalArgs a) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += gridDim.x * kBlock) {
    const uint32_t c = a.seq[i];
    auto pos_of = [&](uint32_t corner, int32_t (&o)[3]) { const int32_t* q = a.pos_by_vertex + (size_t)a.c2v_pos[corner] * 3; o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; };
    int32_t Pc[3];
    pos_of(c, Pc);
    // the fan of the vertex on THIS attribute's table: swing left to its start, then right over every face (the reference's walk)
    uint32_t cur = c;
    for (uint32_t guard = 0; guard < (1u << 24); ++guard) { const uint32_t o = a.opp[cnext(cur)]; if (o == kNoneD) break; cur = cnext(o); if (cur == c) break; }
    const uint32_t start = cur;
    int64_t sum[3] = {0, 0, 0};
    for (uint32_t guard = 0; guard < (1u << 24); ++guard) {
      int32_t Pn[3], Pp[3];
      pos_of(cnext(cur), Pn); pos_of(cprev(cur), Pp);
      add_face_normal<false>(Pn, Pp, Pc, sum);
      const uint32_t o = a.opp[cprev(cur)];
      if (o == kNoneD) break;
      cur = cprev(o);
      if (cur == start) break;
    }
    const int64_t upper = 1ll << 29;
    const int64_t abs_sum = wadd64(wadd64(wabs64(sum[0]), wabs64(sum[1])), wabs64(sum[2]));
    if (abs_sum > upper) { const int64_t quot = abs_sum / upper; sum[0] = wdiv64(sum[0], quot); sum[1] = wdiv64(sum[1], quot); sum[2] = wdiv64(sum[2], quot); }
    const int32_t n0 = (int32_t)sum[0], n1 = (int32_t)sum[1], n2 = (int32_t)sum[2];
    int32_t p0 = 0, p1 = 0;
    if (!(n0 == 0 && n1 == 0 && n2 == 0)) oct_quantize((float)n0, (float)n1, (float)n2, p0, p1);
    if (a.flips[i]) { p0 = wmul(p0, -1); p1 = wmul(p1, -1); }   // the encoder's choice (Q8), read back from its rABS stream
    int32_t o0, o1;
    oct_orthogonal_inverse(p0, p1, (int32_t)a.sym[(size_t)i * 2], (int32_t)a.sym[(size_t)i * 2 + 1], o0, o1);
    int32_t* dst = a.oct_by_vertex + (size_t)a.c2v_att[c] * 2;
    dst[0] = o0; dst[1] = o1;
  }
}
__global__ __launch_bounds__(kBlock) void k_last_corners(const uint32_t* __restrict__ c2p, uint64_t corners, uint32_t* __restrict__ last) {
  for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < corners; c += (uint64_t)gridDim.x * kBlock) atomicMax(&last[c2p[c]], (uint32_t)c + 1u);
}
__global__ __launch_bounds__(kBlock) void k_dequantize(DequantizeArgs a) {
  for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < a.corners; c += (uint64_t)gridDim.x * kBlock) {
    const uint32_t p = a.c2p[c], v = a.c2v[c];
    if (a.last_corner && a.last_corner[p] != (uint32_t)c + 1u) continue;   // (one writer per point: its last corner)
    if (a.kind == 2) {          // coordinate-wise
      for (int k = 0; k < a.N; ++k) a.out[(size_t)p * a.N + k] = a.mn[k] + (float)a.q[(size_t)v * a.N + k] * a.delta;
    } else if (a.kind == 3) {   // octahedral, 8 bits
      const float u = (float)a.q[(size_t)v * 2] / 127.0f - 1.0f, w = (float)a.q[(size_t)v * 2 + 1] / 127.0f - 1.0f;
      float x = 1.0f - fabsf(u) - fabsf(w), y = u, z = w;
      if (x < 0.0f) { const float ya = y, za = z; y = (ya < 0.0f ? -1.0f : 1.0f) * (1.0f - fabsf(za)); z = (za < 0.0f ? -1.0f : 1.0f) * (1.0f - fabsf(ya)); }
      const float nrm = sqrtf(x * x + y * y + z * z);
      if (nrm > 0.0f) { x /= nrm; y /= nrm; z /= nrm; }
      a.out[(size_t)p * 3] = x; a.out[(size_t)p * 3 + 1] = y; a.out[(size_t)p * 3 + 2] = z;
    } else {                    // ToBits: the portable values are the values
      for (int k = 0; k < a.N; ++k) a.out[(size_t)p * a.N + k] = __int_as_float(a.q[(size_t)v * a.N + k]);
    }
  }
}

inline uint32_t grid_for(uint64_t work, uint32_t cap = 256 * 8) {
  uint64_t b = (work + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  b = (b + 7) & ~(uint64_t)7;   // DMI_FOR_SEQUENCE needs a multiple of 8 (one slice per XCD)
  return (uint32_t)(b > cap ? cap : b);
}

// ---- kernel wrappers: one work item per launch, or many (the same phase of a whole batch of jobs) ----
// DMI_KERNEL2: the single-item launch and the batch launch run different instantiations; OCC: an occupancy attribute for the single one
#define DMI_KERNEL2(NAME, BODY, BODY_MULTI, ARGS, THREADS, OCC)                                                                       \
  __global__ __launch_bounds__(THREADS) OCC void NAME(ARGS a) { BODY(a, blockIdx.x, gridDim.x); }                                     \
  __global__ __launch_bounds__(THREADS) DMI_MULTI_OCC void NAME##_multi(const ARGS* __restrict__ items, const uint2* __restrict__ block_info,       \
                                                          const uint32_t* __restrict__ item_blocks) {                                 \
    const uint2 bi = block_info[blockIdx.x];                                                                                          \
    BODY_MULTI(items[bi.x], bi.y, item_blocks[bi.x]);                                                                                  \
  }
#ifndef DMI_MULTI_OCC
#define DMI_MULTI_OCC
#endif
#ifdef DMI_SWEEP_NO_GLDS   // (A/B builds: scripts/glds_probe.sh)
#define DMI_SWEEP_STAGED false
#ifndef DMI_SWEEP_OCC
#define DMI_SWEEP_OCC
#endif
#else
#define DMI_SWEEP_STAGED true
#define DMI_SWEEP_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
#define DMI_KERNEL(NAME, BODY, ARGS, THREADS)                                                                                         \
  __global__ __launch_bounds__(THREADS) void NAME(ARGS a) { BODY(a, blockIdx.x, gridDim.x); }                                         \
  __global__ __launch_bounds__(THREADS) void NAME##_multi(const ARGS* __restrict__ items, const uint2* __restrict__ block_info,       \
                                                          const uint32_t* __restrict__ item_blocks) {                                 \
    const uint2 bi = block_info[blockIdx.x];                                                                                          \
    BODY(items[bi.x], bi.y, item_blocks[bi.x]);                                                                                        \
  }
DMI_KERNEL(k_value_ranges, k_value_ranges_body, RangeArgs, kBlock)
DMI_KERNEL(k_value_ranges_final, k_value_ranges_final_body, RangeArgs, kBlock)
// k_seq_quantize at the compiler's own choice: 104 SGPRs = 7 waves per SIMD (gfx9 SGPR file: 800 per SIMD); capped to 8 waves' worth (73 SGPRs
// parked in VGPR lanes, still 55 VGPRs) the 10M-triangle gather takes 91.5 µs instead of 97.3
#ifndef DMI_SEQ_OCC
#define DMI_SEQ_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
DMI_KERNEL2(k_seq_quantize, k_seq_quantize_body<DMI_KTILE_SINGLE>, k_seq_quantize_body<DMI_KTILE_MULTI>, SeqQuantArgs, kBlock, DMI_SEQ_OCC)
// … and sequences above kSeqQuantizeBigEntries = 2^24 (a 100M-triangle mesh: the gather is bound by what survives in L2 between rings, and fewer
// lines in flight keep more of it) take two entries per thread at the compiler's own 7 waves: 100M triangles 1.75–1.93 ms against 2.27
DMI_KERNEL(k_seq_quantize_big, k_seq_quantize_body<2>, SeqQuantArgs, kBlock)
DMI_KERNEL(k_i32_minmax_final, k_i32_minmax_final_body, MinMaxArgs, kBlock)
DMI_KERNEL(k_predict_fused_pnu, (k_predict_fused_body<true, true, true, false>), FusedArgs, kBlock)
DMI_KERNEL(k_predict_fused_pn, (k_predict_fused_body<true, true, false, false>), FusedArgs, kBlock)
DMI_KERNEL(k_predict_fused_pu, (k_predict_fused_body<true, false, true, false>), FusedArgs, kBlock)
DMI_KERNEL(k_predict_fused_n, (k_predict_fused_body<false, true, false, false>), FusedArgs, kBlock)
DMI_KERNEL2(k_predict_packed_pnu, (k_predict_fused_body<true, true, true, true, DMI_SWEEP_STAGED>), (k_predict_fused_body<true, true, true, true>), FusedArgs, kBlock, DMI_SWEEP_OCC)
DMI_KERNEL2(k_predict_packed_pn, (k_predict_fused_body<true, true, false, true, DMI_SWEEP_STAGED>), (k_predict_fused_body<true, true, false, true>), FusedArgs, kBlock, DMI_SWEEP_OCC)
DMI_KERNEL2(k_predict_packed_pu, (k_predict_fused_body<true, false, true, true, DMI_SWEEP_STAGED>), (k_predict_fused_body<true, false, true, true>), FusedArgs, kBlock, DMI_SWEEP_OCC)
DMI_KERNEL(k_predict_packed_n, (k_predict_fused_body<false, true, false, true>), FusedArgs, kBlock)
DMI_KERNEL(k_predict_window_pnu, (k_predict_window_body<true, true>), FusedArgs, kBlock)
DMI_KERNEL(k_predict_window_pn, (k_predict_window_body<true, false>), FusedArgs, kBlock)
DMI_KERNEL(k_predict_window_pu, (k_predict_window_body<false, true>), FusedArgs, kBlock)
DMI_KERNEL(k_pred_parallelogram_wrapped1, k_pred_parallelogram_wrapped_body<1>, ParArgs, kBlock)
DMI_KERNEL(k_pred_parallelogram_wrapped2, k_pred_parallelogram_wrapped_body<2>, ParArgs, kBlock)
DMI_KERNEL(k_pred_parallelogram_wrapped3, k_pred_parallelogram_wrapped_body<3>, ParArgs, kBlock)
DMI_KERNEL(k_pred_parallelogram_wrapped4, k_pred_parallelogram_wrapped_body<4>, ParArgs, kBlock)
DMI_KERNEL(k_pred_delta_difference, k_pred_delta_difference_body, DeltaArgs, kBlock)
DMI_KERNEL(k_pred_texcoord_wrapped, k_pred_texcoord_wrapped_body, TexArgs, kBlock)
DMI_KERNEL(k_orient_summary, k_orient_summary_body, OrientArgs, 64)
DMI_KERNEL(k_histogram, k_histogram_body, HistArgs, kBlock)
DMI_KERNEL(k_texcoord_fixup, k_texcoord_fixup_body, FusedArgs, kBlock)

thread_local std::vector<KernelStep>* g_step_sink = nullptr;

template <class Args>
void emit(int id, int level, const Args& a, uint32_t blocks, uint32_t lds, hipStream_t s) {
  static_assert(sizeof(Args) <= sizeof(KernelStep::args), "KernelStep::args too small");
  if (blocks == 0) return;
  KernelStep st{};
  st.id = id; st.level = level; st.blocks = blocks; st.lds = lds; st.args_size = (uint32_t)sizeof(Args);
  std::memcpy(st.args, &a, sizeof(Args));
  if (g_step_sink) g_step_sink->push_back(st);
  else launch_step(st, s);
}

}  // namespace

void set_step_sink(std::vector<KernelStep>* sink) { g_step_sink = sink; }
bool step_sink_push(const KernelStep& st) { if (!g_step_sink) return false; g_step_sink->push_back(st); return true; }
bool step_sink_active() { return g_step_sink != nullptr; }

#define DMI_CASE(ID, NAME, ARGS, THREADS) \
  case ID: hipLaunchKernelGGL(NAME, st.blocks, THREADS, st.lds, s, *reinterpret_cast<const ARGS*>(st.args)); break;
#define DMI_CASE_MULTI(ID, NAME, ARGS, THREADS) \
  case ID: hipLaunchKernelGGL(NAME##_multi, total_blocks, THREADS, lds, s, static_cast<const ARGS*>(items), block_info, item_blocks); break;
#define DMI_ALL_KERNELS(X)                                                   \
  X(K_RANGES, k_value_ranges, RangeArgs, kBlock)                             \
  X(K_RANGES_FINAL, k_value_ranges_final, RangeArgs, kBlock)                 \
  X(K_SEQ_QUANT, k_seq_quantize, SeqQuantArgs, kBlock)                       \
  X(K_SEQ_QUANT_BIG, k_seq_quantize_big, SeqQuantArgs, kBlock)               \
  X(K_I32_FINAL, k_i32_minmax_final, MinMaxArgs, kBlock)                     \
  X(K_FUSED_PNU, k_predict_fused_pnu, FusedArgs, kBlock)                     \
  X(K_FUSED_PN, k_predict_fused_pn, FusedArgs, kBlock)                       \
  X(K_FUSED_PU, k_predict_fused_pu, FusedArgs, kBlock)                       \
  X(K_FUSED_N, k_predict_fused_n, FusedArgs, kBlock)                         \
  X(K_PACKED_PNU, k_predict_packed_pnu, FusedArgs, kBlock)                   \
  X(K_PACKED_PN, k_predict_packed_pn, FusedArgs, kBlock)                     \
  X(K_PACKED_PU, k_predict_packed_pu, FusedArgs, kBlock)                     \
  X(K_PACKED_N, k_predict_packed_n, FusedArgs, kBlock)                       \
  X(K_WINDOW_PNU, k_predict_window_pnu, FusedArgs, kBlock)                   \
  X(K_WINDOW_PN, k_predict_window_pn, FusedArgs, kBlock)                     \
  X(K_WINDOW_PU, k_predict_window_pu, FusedArgs, kBlock)                     \
  X(K_PAR1, k_pred_parallelogram_wrapped1, ParArgs, kBlock)                  \
  X(K_PAR2, k_pred_parallelogram_wrapped2, ParArgs, kBlock)                  \
  X(K_PAR3, k_pred_parallelogram_wrapped3, ParArgs, kBlock)                  \
  X(K_PAR4, k_pred_parallelogram_wrapped4, ParArgs, kBlock)                  \
  X(K_DELTA, k_pred_delta_difference, DeltaArgs, kBlock)                     \
  X(K_TEX, k_pred_texcoord_wrapped, TexArgs, kBlock)                         \
  X(K_ORIENT, k_orient_summary, OrientArgs, 64)                              \
  X(K_HIST, k_histogram, HistArgs, kBlock)                                   \
  X(K_TEX_FIXUP, k_texcoord_fixup, FusedArgs, kBlock)

void launch_step(const KernelStep& st, hipStream_t s) {
  switch (st.id) {
    DMI_ALL_KERNELS(DMI_CASE)
    default: launch_prep_step(st, s); break;
  }
}
void launch_steps_multi(int id, const void* items, const uint2* block_info, const uint32_t* item_blocks, uint32_t total_blocks, uint32_t lds, hipStream_t s) {
  if (!total_blocks) return;
  switch (id) {
    DMI_ALL_KERNELS(DMI_CASE_MULTI)
    default: launch_prep_steps_multi(id, items, block_info, item_blocks, total_blocks, s); break;
  }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (each emits one KernelStep: launched at once, or collected when a sink is set)
// ------------------------------------------------------------------------------------------------
void launch_value_ranges(RangeArgs& args, hipStream_t s) {
  if (args.count == 0) return;
  uint32_t total = 0;
  for (int i = 0; i < args.count; ++i) {
    RangeAtt& a = args.a[i];
    a.blocks = a.kind == 2 ? 0u : std::min<uint32_t>(kRangeMaxBlocks, std::max<uint32_t>(1u, (a.n + kBlock - 1) / kBlock));
    a.first_block = total;
    total += a.blocks;
  }
  emit(K_RANGES, 0, args, total, 0, s);
  emit(K_RANGES_FINAL, 1, args, (uint32_t)args.count, 0, s);
}

void launch_value_range_partials(RangeArgs& args, uint32_t max_blocks, hipStream_t s) {
  uint32_t total = 0;
  for (int i = 0; i < args.count; ++i) {
    RangeAtt& a = args.a[i];
    a.blocks = a.kind == 2 ? 0u : std::min<uint32_t>(std::min(max_blocks, kRangeMaxBlocks), std::max<uint32_t>(1u, (a.n + DMI_RANGE_UNROLL * kBlock - 1) / (DMI_RANGE_UNROLL * kBlock)));
    a.first_block = total;
    total += a.blocks;
  }
  if (total) hipLaunchKernelGGL(k_value_ranges, total, kBlock, 0, s, args);
}

void launch_i32_minmax_final(const MinMaxArgs& args, hipStream_t s) { emit(K_I32_FINAL, 3, args, (uint32_t)args.count, 0, s); }
// (launched directly: the early stage exists for single one-shot jobs only, never under a batch's step sink)
uint32_t value_quantize_rec_blocks(uint32_t n) { return grid_for(((uint64_t)n + kRecPer - 1) / kRecPer, 2048); }   // (4096 blocks: 47.6 µs, 8192: 49.9 against 45.4 — every block folds the range partials)   // = partial pairs written per attribute
void launch_value_quantize_rec(const ValueRecArgs& a, hipStream_t s) {
  if (!a.n) return;
  const uint32_t g = value_quantize_rec_blocks(a.n);
  if (a.nrm && a.uv) hipLaunchKernelGGL(k_value_quantize_rec_pnu, g, kBlock, 0, s, a);
  else if (a.nrm) hipLaunchKernelGGL(k_value_quantize_rec_pn, g, kBlock, 0, s, a);
  else hipLaunchKernelGGL(k_value_quantize_rec_pu, g, kBlock, 0, s, a);
}
void launch_seq_gather_rec(const GatherRecArgs& g, hipStream_t s) {
  // one tile per block (more entries per thread, non-temporal loads and stores, capped grids: 44.6–47.9 µs all of them on the 10M workload — the gather is
  // bound by what a permutation of 16-byte records gets out of HBM, 170 MB at 3.7 TB/s)
  if (g.n) hipLaunchKernelGGL(k_seq_gather_rec, grid_for(((uint64_t)g.n + kGatherPer - 1) / kGatherPer, kSeqQuantizeMaxBlocks), kBlock, 0, s, g);
}

// grid of k_seq_quantize (a DMI_FOR_TILES kernel: any grid is correct, this one gives every block work): the steps of a batch are recorded
// (step sink) and run as the `_multi` launch, two entries per thread
// (DMI_SEQ_BIG_ENTRIES: the length above which the big form runs — read per call so that the tests can put small meshes through it)
static uint32_t seq_big_entries() { return dbg().seq_big_entries ? dbg().seq_big_entries : kSeqQuantizeBigEntries; }
uint32_t seq_quantize_blocks(uint32_t n) {
  const uint64_t kt = (step_sink_active() || n > seq_big_entries()) ? DMI_KTILE_MULTI : DMI_KTILE_SINGLE;
  return grid_for(((uint64_t)n + kt - 1) / kt, kSeqQuantizeMaxBlocks);
}
void launch_seq_quantize(const uint32_t* s2p, const uint32_t* dest, uint32_t n, const QuantArgs& args, hipStream_t s) {
  SeqQuantArgs sq{};
  sq.s2p = s2p; sq.dest = dest; sq.n = n; sq.q = args;
  emit(n > seq_big_entries() ? K_SEQ_QUANT_BIG : K_SEQ_QUANT, 2, sq, seq_quantize_blocks(n), 0, s);
}

void launch_pred_parallelogram_wrapped(const uint32_t* seq, uint32_t n, const uint32_t* c2r, const uint32_t* opp,
                                       const int32_t* qs, const int32_t* minmax, int N, void* sym, bool sym16, hipStream_t s) {
  ParArgs pa{seq, c2r, opp, qs, minmax, sym, n, sym16 ? 1u : 0u};
  emit(N == 1 ? K_PAR1 : (N == 2 ? K_PAR2 : (N == 3 ? K_PAR3 : K_PAR4)), 4, pa, grid_for(n), 0, s);
}

void launch_pred_delta_difference(uint32_t n, const int32_t* qs, int N, void* sym, bool sym16, hipStream_t s) {
  DeltaArgs da{(uint64_t)n * N, qs, sym, N, sym16 ? 1 : 0};
  emit(K_DELTA, 4, da, da.n_comp ? grid_for(da.n_comp) : 0u, 0, s);
}

void launch_pred_texcoord_wrapped(const uint32_t* seq, uint32_t n, const uint32_t* c2r, const int32_t* qs, const uint32_t* c2r_pos,
                                  const void* qs_pos, int pos_fmt, const int32_t* minmax, void* sym, bool sym16, uint8_t* orient, hipStream_t s) {
  TexArgs ta{seq, c2r, qs, c2r_pos, qs_pos, minmax, sym, orient, n, sym16 ? 1u : 0u, pos_fmt, 0};
  emit(K_TEX, 4, ta, grid_for(((uint64_t)n + kTexTile - 1) / kTexTile), 0, s);
}

void launch_build_fans(const uint32_t* seq, uint32_t n, const uint32_t* c2r, const uint32_t* opp, uint32_t* hdr, uint32_t* apex, uint32_t* fan, bool centre_in_apex,
                       hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_build_fans, grid_for(n, 8192), kBlock, 0, s, seq, n, c2r, opp, hdr, apex, fan, centre_in_apex ? 1 : 0);
}

void launch_build_fans_rec(const uint32_t* seq, uint32_t n, const uint32_t* frec, uint32_t* hdr, uint32_t* apex, uint32_t* fan, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_build_fans_rec, grid_for(n, 8192), kBlock, 0, s, seq, n, reinterpret_cast<const uint4*>(frec), hdr, apex, fan);
}

void launch_build_fans_batch(const FanItem* items_dev, uint32_t n_items, uint32_t total, hipStream_t s) {
  if (n_items && total) hipLaunchKernelGGL(k_build_fans_batch, grid_for(total, 65535u * 16u), kBlock, 0, s, items_dev, n_items, total);
}

uint32_t predict_fused_blocks(uint32_t n) {
  const uint32_t env_cap = dbg().fused_grid;   // tuning aid
  return grid_for(n, env_cap ? std::min(env_cap, kSweepMaxBlocks) : 8192u);   // 2-3 chunks per block: measured best on the 10M workload
}
void launch_predict_fused(const FusedArgs& a, hipStream_t s) {
  if (a.n == 0) return;
  const uint32_t g = predict_fused_blocks(a.n);
  int id = !a.sym_pos ? K_FUSED_N /* a normal attribute on its own table */ : ((a.qs_nrm && a.qs_uv) ? K_FUSED_PNU : (a.qs_nrm ? K_FUSED_PN : K_FUSED_PU));
  if (a.packed) id += K_PACKED_PNU - K_FUSED_PNU;
  const bool windows = dbg_on(DMI_DBG_FUSED_WINDOWS);   // LDS-staged neighbourhoods (see k_predict_window_body)
  if (windows && a.packed && a.sym_pos && a.face_stride != 8u) id = id == K_PACKED_PNU ? K_WINDOW_PNU : (id == K_PACKED_PN ? K_WINDOW_PN : K_WINDOW_PU);
  const uint32_t env_lds = dbg().fused_lds;   // tuning aid: unused dynamic LDS per block = fewer blocks per CU
  emit(id, 4, a, g, env_lds, s);
  if (a.qs_uv && a.sym_pos && id != K_WINDOW_PNU && id != K_WINDOW_PU) emit(K_TEX_FIXUP, 5, a, 64u, 0, s);   // the entries the sweep deferred (usually none: the launch finds count = 0)
}

void launch_decode_normals(const DecodeNormalArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_decode_normals, grid_for(a.n, 8192), kBlock, 0, s, a); }
void launch_last_corners(const uint32_t* c2p, uint64_t corners, uint32_t* last_corner, hipStream_t s) { if (corners) hipLaunchKernelGGL(k_last_corners, grid_for(corners, 8192), kBlock, 0, s, c2p, corners, last_corner); }
void launch_dequantize(const DequantizeArgs& a, hipStream_t s) { if (a.corners) hipLaunchKernelGGL(k_dequantize, grid_for(a.corners, 8192), kBlock, 0, s, a); }

uint32_t orient_summary_blocks(uint32_t n) { return (n + kOrientChunk - 1) / kOrientChunk; }
void launch_orient_summary(const uint8_t* orient, uint32_t n, uint32_t* summary, uint32_t*, hipStream_t s) {
  OrientArgs oa{orient, summary, n, 0u};
  emit(K_ORIENT, 5, oa, orient_summary_blocks(n), 0, s);
}

void launch_histograms(HistArgs& args, hipStream_t s) {
  uint32_t total = 0;
  size_t lds = 0;
  for (int i = 0; i < args.count; ++i) {
    HistAtt& a = args.a[i];
    // a block flushes its private bins with one global atomic per occupied bin: it should count many more symbols than it has bins
    // (a 60k-symbol attribute of a small mesh on 240 blocks paid 8 flushed bins per symbol)
    a.blocks = a.n ? (uint32_t)std::min<uint64_t>(512u, std::max<uint64_t>(1u, (a.n + 16383u) / 16384u)) : 0u;
    a.first_block = total;
    total += a.blocks;
    if (a.blocks) lds = std::max(lds, ((size_t)std::min(a.bins, kLdsBins) + kHistHotWords) * 4);
  }
  args.hist_blocks = total;
  const uint32_t orient_blocks = args.orient.orient ? orient_summary_blocks(args.orient.n) : 0u;
  emit(K_HIST, 6, args, total + orient_blocks, (uint32_t)lds, s);
}

}  // namespace dmi
