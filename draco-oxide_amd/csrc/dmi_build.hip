// dmi_build.hip — MeshBuilder::build on the device (gfx950, wave64), ONE launch per kernel for all meshes of a batch.
//
// Reference stages (paths relative to draco-oxide/src/):
//   value dedup     core/attribute/mod.rs:394-452   Attribute::from — pairwise `==`, duplicates map to their FIRST occurrence, ids compacted
//                                                   in first-occurrence order; f32 `==` classes (-0.0 == 0.0, a row holding a NaN equals
//                                                   nothing: macros/.../lib.rs:167-175)
//   point merge     core/mesh/builder.rs:194-279    points whose unique values are BYTE-identical in every attribute are one point (first
//                                                   occurrence kept): the value id for a row that equals itself, the row's own bytes for a row
//                                                   holding a NaN (hash_vertex :254-279 hashes bytes — two byte-identical NaN rows merge)
//   degenerate      core/mesh/builder.rs:77-79      faces with a repeated point id are dropped
//   unused points   core/mesh/builder.rs:129-189    points no face references are removed, faces renumbered; a value that loses its last
//                                                   point leaves the buffer (Attribute::remove, mod.rs:454-483), order preserved
// The reference's builders are O(V²) scans; what they compute is order statistics of equivalence classes: "first occurrence" = the
// smallest index of a class, "rank" = a prefix sum over first-occurrence flags.  Here every class is found with an open-addressing hash
// table whose slots hold a ROW INDEX: inserting row p either claims an empty slot, or meets a slot whose row compares equal — then the slot
// keeps the smaller of the two indices (atomicMin) — or probes on.  Whatever the interleaving, a slot ends up holding the smallest index of
// its class, so the result does not depend on scheduling.  Ranks are exclusive scans; compactions are scatters at scanned offsets.
// Rows are read where the upload left them: tightly packed (the host packed them) or with their accessor's stride (the caller's buffer came up as
// it is: dmi_build.cpp "in place").
// Covered class: every attribute of a mesh has the same point count, rows of 1–4 four-byte components, faces index existing points, at least
// one face survives.  Anything else is FLAGGED and the host builder (host_mesh.cpp) takes that mesh — nothing is approximated here.
#include "dmi_device.hpp"
#include <algorithm>

namespace dmi {
namespace {

constexpr int kBlock = 256;
constexpr uint32_t kNoneD = 0xFFFFFFFFu;
// A table is at most half full, so a probe sequence of honest data is a handful of slots; rows chosen to collide under `mix` would make an
// insert walk them all (quadratic: a kernel that does not come back).  Past this many probes the mesh is flagged and the host builder takes it.
constexpr uint32_t kMaxProbes = 2048;

inline uint32_t grid_of(uint64_t n) {
  const uint64_t g = (n + kBlock - 1) / kBlock;
  return (uint32_t)(g > 65535ull * 32 ? 65535ull * 32 : (g ? g : 1));
}
// the last entry whose offset is ≤ x (offsets ascending)
template <class Get>
__device__ __forceinline__ uint32_t find_last(uint32_t n, uint32_t x, Get off) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off(mid) <= x) lo = mid; else hi = mid; }
  return lo;
}
__device__ __forceinline__ uint32_t mesh_of_face(const MbArgs& a, uint32_t f) { return a.M == 1 ? 0u : find_last(a.M, f, [&](uint32_t k) { return a.meshes[k].face_off; }); }
__device__ __forceinline__ uint32_t mesh_of_point(const MbArgs& a, uint32_t p) { return a.M == 1 ? 0u : find_last(a.M, p, [&](uint32_t k) { return a.meshes[k].point_off; }); }
__device__ __forceinline__ uint32_t item_of(const MbArgs& a, uint32_t ap) { return a.n_items == 1 ? 0u : find_last(a.n_items, ap, [&](uint32_t k) { return a.items[k].ap_off; }); }

__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t u = __shfl_up(v, off, 64); if ((int)(threadIdx.x & 63) >= off) v += u; }
  return v;
}
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[kBlock / 64 + 1];
  const uint32_t inc = wave_inclusive(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) { if (w < wave) base += wsum[w]; all += wsum[w]; }
  if (total) *total = all;
  return base + inc - v;
}
__device__ __forceinline__ void raise(MbMeshOut* out, uint32_t m, uint32_t bit) {
  if (!(__atomic_load_n(&out[m].flags, __ATOMIC_RELAXED) & bit)) atomicOr(&out[m].flags, bit);
}
__device__ __forceinline__ uint32_t mix(uint32_t h, uint32_t w) {
  h ^= w;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  return h ^ (h >> 16);
}
// `==` on f32 bit patterns once NaN rows are out of the way: equal bits, or both zeros (-0.0 == 0.0) — a zero of either sign becomes +0
__device__ __forceinline__ uint32_t canon(uint32_t w, bool is_float) { return (is_float && (w << 1) == 0u) ? 0u : w; }
__device__ __forceinline__ bool is_nan_bits(uint32_t w) { return (w & 0x7FFFFFFFu) > 0x7F800000u; }

// ---- faces: index range, largest referenced point per mesh (num_vertices = max + 1, builder.rs:196-199) ----
// A block owns kRangePer·kBlock consecutive corners; a thread keeps the maximum of the mesh it is in and hands it over when the mesh
// changes (rare) or at the end, where a wavefront whose lanes all sit in one mesh sends ONE atomic (same-address atomics execute one after
// the other, ≈ 11 ns each: one per corner made this kernel 20–44 ms).
constexpr int kRangePer = 32;
__device__ __forceinline__ void nv_max(MbMeshOut* out, uint32_t m, uint32_t v) {
  if (v && __atomic_load_n(&out[m].nv, __ATOMIC_RELAXED) < v) atomicMax(&out[m].nv, v);
}
__global__ __launch_bounds__(kBlock) void k_mb_face_range(const MbArgs a) {
  const uint64_t C = 3ull * a.total_faces, base = (uint64_t)blockIdx.x * (kRangePer * kBlock);
  uint32_t cur_m = kNoneD, cur_max = 0, cur_end = 0, cur_P = 0;   // cur_end: first face past mesh cur_m
  for (int k = 0; k < kRangePer; ++k) {
    const uint64_t c = base + (uint64_t)k * kBlock + threadIdx.x;
    if (c >= C) break;
    const uint32_t f = (uint32_t)(c / 3u);
    if (cur_m == kNoneD || f >= cur_end) {
      if (cur_m != kNoneD) nv_max(a.mesh_out, cur_m, cur_max);
      cur_m = mesh_of_face(a, f);
      cur_end = a.meshes[cur_m].face_off + a.meshes[cur_m].F;
      cur_P = a.meshes[cur_m].P;
      cur_max = 0;
    }
    const uint32_t p = a.raw_faces[c];
    if (p >= cur_P) raise(a.mesh_out, cur_m, MB_BAD_INDEX);
    else cur_max = max(cur_max, p + 1);
  }
  const bool have = cur_m != kNoneD;
  const uint64_t mask = __ballot(have);
  if (mask == 0ull) return;
  const uint32_t m0 = __shfl(cur_m, __ffsll((long long)mask) - 1, 64);
  if (__ballot(have && cur_m != m0) == 0ull) {
    uint32_t x = have ? cur_max : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = max(x, (uint32_t)__shfl_down(x, off, 64));
    if ((threadIdx.x & 63) == 0) nv_max(a.mesh_out, m0, x);
  } else if (have) {
    nv_max(a.mesh_out, cur_m, cur_max);
  }
}

// ---- value classes per attribute ----
// The table finds BYTE classes (what the point merge compares, builder.rs:254-279): rows without a NaN by `==` on canonical words — the value
// classes of Attribute::from —, rows holding a NaN by their raw bits (a NaN row is a value of its own whatever the table says, but its point
// merges with the point of a byte-identical NaN row).  A NaN row never compares equal to a NaN-free one under either rule, so one table holds both.
__global__ __launch_bounds__(kBlock) void k_mb_value_insert(const MbArgs a) {
  for (uint32_t ap = blockIdx.x * kBlock + threadIdx.x; ap < a.total_ap; ap += gridDim.x * kBlock) {
    const MbItem it = a.items[item_of(a, ap)];
    const uint32_t p = ap - it.ap_off;
    const uint32_t* __restrict__ rows = a.raw_values + it.row_off;
    const bool fl = it.is_float != 0;
    uint32_t raw[4] = {0u, 0u, 0u, 0u}, w[4] = {0u, 0u, 0u, 0u};
    bool nan = false;
    for (uint32_t k = 0; k < it.words; ++k) {
      raw[k] = rows[(size_t)p * it.stride + k];
      nan = nan || (fl && is_nan_bits(raw[k]));
    }
    const bool cn = fl && !nan;   // compare canonical words (-0.0 == 0.0); a NaN row: its bits as they are
    uint32_t h = nan ? 0x7F4A7C15u : 0x9E3779B9u;
    for (uint32_t k = 0; k < it.words; ++k) { w[k] = canon(raw[k], cn); h = mix(h, w[k]); }
    a.vflag[ap] = nan ? 1u : 0u;   // (parked for k_mb_value_first, which writes the first-occurrence flag here)
    uint32_t* __restrict__ tab = a.vtab + it.tab_off;
    h &= it.tab_mask;
    uint32_t probes = 0;
    for (;; ++probes) {
      if (probes > kMaxProbes) { raise(a.mesh_out, it.mesh, MB_CROWDED); break; }   // rows crafted to collide: the host builder takes the mesh
      // (claim first: most rows of most meshes are the first of their class and find an empty slot — one memory-side operation instead of a load and a claim;
      //  a failed claim returns the occupant the load would have returned)
      const uint32_t s = atomicCAS(&tab[h], kNoneD, p);
      if (s == kNoneD) break;
      bool same = true;   // (s may be lowered meanwhile by another member of ITS class: every index a slot ever holds is of one class)
      for (uint32_t k = 0; k < it.words; ++k) same = same && canon(rows[(size_t)s * it.stride + k], cn) == w[k];
      if (same) { if (p < s) atomicMin(&tab[h], p); break; }
      h = (h + 1) & it.tab_mask;
    }
    a.vslot[ap] = probes > kMaxProbes ? kNoneD : h;
  }
}
// representative (first occurrence) of every row as a VALUE — a NaN row is its own — and as a BYTE CLASS (parked in vslot, whose slot number
// nobody needs any more: the point merge's key); first-occurrence flags for the rank scan
__global__ __launch_bounds__(kBlock) void k_mb_value_first(const MbArgs a) {
  for (uint32_t ap = blockIdx.x * kBlock + threadIdx.x; ap < a.total_ap; ap += gridDim.x * kBlock) {
    const MbItem& it = a.items[item_of(a, ap)];
    const uint32_t p = ap - it.ap_off, s = a.vslot[ap];
    const bool nan = a.vflag[ap] != 0u;
    const uint32_t cls = s == kNoneD ? p : a.vtab[it.tab_off + s];
    const uint32_t rep = nan ? p : cls;
    a.vslot[ap] = cls;
    a.vid[ap] = rep;
    a.vflag[ap] = rep == p ? 1u : 0u;
  }
}
// value id = rank of the representative among the first occurrences (mod.rs:426-443); vfirst[id] = the point that carries the value
__global__ __launch_bounds__(kBlock) void k_mb_value_ids(const MbArgs a) {
  for (uint32_t ap = blockIdx.x * kBlock + threadIdx.x; ap < a.total_ap; ap += gridDim.x * kBlock) {
    const MbItem& it = a.items[item_of(a, ap)];
    const uint32_t p = ap - it.ap_off, rep = a.vid[ap];
    const uint32_t id = a.vflag[it.ap_off + rep] - a.vflag[it.ap_off];
    a.vid[ap] = id;
    if (rep == p) a.vfirst[it.ap_off + id] = p;
  }
}

// ---- point classes: the tuple of byte classes (builder.rs:254-279 hashes the unique values' bytes at the point; vslot holds, per attribute and
// point, the smallest row index of the row's byte class — k_mb_value_first) ----
constexpr int kMaxKey = (int)kMbMaxAtts;   // attributes per mesh the device form takes
__global__ __launch_bounds__(kBlock) void k_mb_point_insert(const MbArgs a) {
  for (uint32_t gp = blockIdx.x * kBlock + threadIdx.x; gp < a.total_points; gp += gridDim.x * kBlock) {
    const MbMesh me = a.meshes[mesh_of_point(a, gp)];
    const uint32_t p = gp - me.point_off;
    a.pslot[gp] = kNoneD;
    if (p >= a.mesh_out[me.index].nv) continue;   // points past the largest referenced one take no part (builder.rs:200, :258)
    uint32_t key[kMaxKey];
    uint32_t h = 0x9E3779B9u;
    for (uint32_t k = 0; k < me.n_items; ++k) { key[k] = a.vslot[a.items[me.item0 + k].ap_off + p]; h = mix(h, key[k]); }
    uint32_t* __restrict__ tab = a.ptab + me.ptab_off;
    h &= me.ptab_mask;
    uint32_t probes = 0;
    for (;; ++probes) {
      if (probes > kMaxProbes) { raise(a.mesh_out, me.index, MB_CROWDED); break; }
      // (claim first: most rows of most meshes are the first of their class and find an empty slot — one memory-side operation instead of a load and a claim;
      //  a failed claim returns the occupant the load would have returned)
      const uint32_t s = atomicCAS(&tab[h], kNoneD, p);
      if (s == kNoneD) break;
      bool same = true;
      for (uint32_t k = 0; k < me.n_items; ++k) same = same && a.vslot[a.items[me.item0 + k].ap_off + s] == key[k];
      if (same) { if (p < s) atomicMin(&tab[h], p); break; }
      h = (h + 1) & me.ptab_mask;
    }
    a.pslot[gp] = probes > kMaxProbes ? kNoneD : h;
  }
}
__global__ __launch_bounds__(kBlock) void k_mb_point_first(const MbArgs a) {
  for (uint32_t gp = blockIdx.x * kBlock + threadIdx.x; gp < a.total_points; gp += gridDim.x * kBlock) {
    const MbMesh& me = a.meshes[mesh_of_point(a, gp)];
    const uint32_t p = gp - me.point_off, s = a.pslot[gp];
    const uint32_t rep = s == kNoneD ? kNoneD : a.ptab[me.ptab_off + s];
    a.prep[gp] = rep;
    a.pflag[gp] = rep == p ? 1u : 0u;
  }
}
// faces through the point map; degenerate ones dropped; the merged points a surviving face references are marked
__global__ __launch_bounds__(kBlock) void k_mb_faces_map(const MbArgs a) {
  for (uint32_t f = blockIdx.x * kBlock + threadIdx.x; f < a.total_faces; f += gridDim.x * kBlock) {
    const MbMesh& me = a.meshes[mesh_of_face(a, f)];
    uint32_t v[3];
    bool ok = !(a.mesh_out[me.index].flags & (MB_BAD_INDEX | MB_CROWDED));   // (a flagged mesh keeps no face: the host builder takes it)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t p = a.raw_faces[3ull * f + k];
      v[k] = 0;
      if (!ok || p >= me.P) { ok = false; continue; }
      v[k] = a.pflag[me.point_off + a.prep[me.point_off + p]] - a.pflag[me.point_off];   // rank of the class = the merged point's id
    }
    ok = ok && v[0] != v[1] && v[1] != v[2] && v[0] != v[2];
    a.keep[f] = ok ? 1u : 0u;
    if (!ok) continue;
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.tmp_faces[3ull * f + k] = v[k]; a.used[me.point_off + v[k]] = 1u; }
  }
}
// (after the scans of pflag / used) does point p of its mesh survive, and as which point
__device__ __forceinline__ bool survives(const MbArgs& a, const MbMesh& me, uint32_t p, uint32_t* new_point) {
  if (p >= a.mesh_out[me.index].nv || a.prep[me.point_off + p] != p) return false;
  const uint32_t merged = a.pflag[me.point_off + p] - a.pflag[me.point_off];
  const uint32_t lo = a.used[me.point_off + merged], hi = a.used[me.point_off + merged + 1];
  *new_point = lo - a.used[me.point_off];
  return hi != lo;
}
// values that keep at least one point
__global__ __launch_bounds__(kBlock) void k_mb_values_used(const MbArgs a) {
  for (uint32_t ap = blockIdx.x * kBlock + threadIdx.x; ap < a.total_ap; ap += gridDim.x * kBlock) {
    const MbItem& it = a.items[item_of(a, ap)];
    const MbMesh& me = a.meshes[it.mesh];
    uint32_t q;
    if (survives(a, me, ap - it.ap_off, &q)) a.vused[it.ap_off + a.vid[ap]] = 1u;
  }
}
// counts and output offsets of every mesh / attribute: arena A = faces of all meshes (one array, mesh after mesh) then the point → value
// maps, arena B = the unique values.  One block.
__global__ __launch_bounds__(kBlock) void k_mb_layout(const MbArgs a) {
  for (uint32_t m = threadIdx.x; m < a.M; m += kBlock) {
    const MbMesh& me = a.meshes[m];
    MbMeshOut& o = a.mesh_out[m];
    o.face_out_off = a.keep[me.face_off];
    o.F_out = a.keep[me.face_off + me.F] - a.keep[me.face_off];
    o.P_out = a.used[me.point_off + me.P] - a.used[me.point_off];
    o.classes = a.pflag[me.point_off + me.P] - a.pflag[me.point_off];
    if (o.F_out == 0) o.flags |= MB_EMPTY;
  }
  __syncthreads();
  const uint32_t faces_words = 3u * a.keep[a.total_faces];
  const uint32_t per = (a.n_items + kBlock - 1) / kBlock;
  const uint32_t lo = min(a.n_items, threadIdx.x * per), hi = min(a.n_items, lo + per);
  uint32_t map_sum = 0, val_sum = 0;
  for (uint32_t i = lo; i < hi; ++i) {
    const MbItem& it = a.items[i];
    MbItemOut& o = a.item_out[i];
    o.n_first = a.vflag[it.ap_off + it.P] - a.vflag[it.ap_off];
    o.n_out = a.vused[it.ap_off + it.P] - a.vused[it.ap_off];
    o.has_map = o.n_first != it.P ? 1u : 0u;   // a map exists iff a duplicate was found (mod.rs:444-446); removals keep it
    if (o.has_map) map_sum += (a.mesh_out[it.mesh].P_out + 63u) & ~63u;
    val_sum += (o.n_out * it.words + 63u) & ~63u;
  }
  uint32_t map_total, val_total;
  uint32_t map_run = block_exclusive(map_sum, &map_total) + ((faces_words + 63u) & ~63u);
  uint32_t val_run = block_exclusive(val_sum, &val_total);
  for (uint32_t i = lo; i < hi; ++i) {
    const MbItem& it = a.items[i];
    MbItemOut& o = a.item_out[i];
    o.map_off = map_run;
    o.val_off = val_run;
    if (o.has_map) map_run += (a.mesh_out[it.mesh].P_out + 63u) & ~63u;
    val_run += (o.n_out * it.words + 63u) & ~63u;
  }
  if (threadIdx.x == 0) { a.totals[0] = ((faces_words + 63u) & ~63u) + map_total; a.totals[1] = val_total; a.totals[2] = faces_words; }
}
// surviving faces, renumbered by the surviving points, into arena A
__global__ __launch_bounds__(kBlock) void k_mb_faces_out(const MbArgs a) {
  for (uint32_t f = blockIdx.x * kBlock + threadIdx.x; f < a.total_faces; f += gridDim.x * kBlock) {
    const uint32_t at = a.keep[f];
    if (a.keep[f + 1] == at) continue;
    const MbMesh& me = a.meshes[mesh_of_face(a, f)];
    const uint32_t base = a.used[me.point_off];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.arena_a[3ull * at + k] = a.used[me.point_off + a.tmp_faces[3ull * f + k]] - base;
  }
}
// maps of the surviving points and the values that are left, in first-occurrence order
__global__ __launch_bounds__(kBlock) void k_mb_attributes_out(const MbArgs a) {
  for (uint32_t ap = blockIdx.x * kBlock + threadIdx.x; ap < a.total_ap; ap += gridDim.x * kBlock) {
    const uint32_t i = item_of(a, ap);
    const MbItem& it = a.items[i];
    const MbItemOut& o = a.item_out[i];
    const MbMesh& me = a.meshes[it.mesh];
    const uint32_t p = ap - it.ap_off;
    uint32_t q;
    if (o.has_map && survives(a, me, p, &q)) a.arena_a[o.map_off + q] = a.vused[it.ap_off + a.vid[ap]] - a.vused[it.ap_off];
    // the same index as a VALUE id of this attribute
    if (p < o.n_first && a.vused[ap + 1] != a.vused[ap]) {
      const uint32_t r = a.vused[ap] - a.vused[it.ap_off];
      const uint32_t* __restrict__ src = a.raw_values + it.row_off + (size_t)a.vfirst[ap] * it.stride;
      uint32_t* __restrict__ dst = a.arena_b + o.val_off + (size_t)r * it.words;
      for (uint32_t k = 0; k < it.words; ++k) dst[k] = src[k];
    }
  }
}
// the index arrays of a group in which some are narrower than 32 bits (glTF UNSIGNED_BYTE / UNSIGNED_SHORT): all of them into ONE u32 face array
__global__ __launch_bounds__(kBlock) void k_mb_widen(const MbWiden* __restrict__ items, uint32_t n_items, uint32_t total, const uint8_t* __restrict__ src, uint32_t* __restrict__ dst) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < total; g += gridDim.x * kBlock) {
    const MbWiden& it = items[find_last(n_items, g, [&](uint32_t k) { return items[k].off; })];
    const uint32_t k = g - it.off;
    const uint8_t* s = src + it.src_byte_off;
    dst[it.dst_off + k] = it.bytes == 1 ? (uint32_t)s[k] : it.bytes == 2 ? (uint32_t)reinterpret_cast<const uint16_t*>(s)[k] : reinterpret_cast<const uint32_t*>(s)[k];
  }
}

}  // namespace

hipError_t mesh_build_clear(const MbArgs& a, size_t vtab_words, size_t ptab_words, hipStream_t s) {
  ClearRanges r{};   // (one launch: eight hipMemsetAsync before)
  r.add(a.vtab, vtab_words * 4, 0xFF); r.add(a.ptab, ptab_words * 4, 0xFF); r.add(a.mesh_out, (size_t)a.M * sizeof(MbMeshOut));
  r.add(a.vflag + a.total_ap, 4); r.add(a.vused, ((size_t)a.total_ap + 1) * 4); r.add(a.pflag + a.total_points, 4);
  r.add(a.used, ((size_t)a.total_points + 1) * 4); r.add(a.keep + a.total_faces, 4);
  launch_clear_ranges(r, s);
  return hipGetLastError();
}
void launch_mesh_build(const MbArgs& a, hipStream_t s) {
  if (!a.M || !a.total_faces || !a.total_ap) return;
  hipLaunchKernelGGL(k_mb_face_range, (uint32_t)((3ull * a.total_faces + kRangePer * kBlock - 1) / (kRangePer * kBlock)), kBlock, 0, s, a);
  hipLaunchKernelGGL(k_mb_value_insert, grid_of(a.total_ap), kBlock, 0, s, a);
  hipLaunchKernelGGL(k_mb_value_first, grid_of(a.total_ap), kBlock, 0, s, a);
  launch_exclusive_scan_u32(a.vflag, a.total_ap + 1, a.scan_partials, s);
  hipLaunchKernelGGL(k_mb_value_ids, grid_of(a.total_ap), kBlock, 0, s, a);
  hipLaunchKernelGGL(k_mb_point_insert, grid_of(a.total_points), kBlock, 0, s, a);
  hipLaunchKernelGGL(k_mb_point_first, grid_of(a.total_points), kBlock, 0, s, a);
  launch_exclusive_scan_u32(a.pflag, a.total_points + 1, a.scan_partials, s);
  hipLaunchKernelGGL(k_mb_faces_map, grid_of(a.total_faces), kBlock, 0, s, a);
  launch_exclusive_scan_u32(a.keep, a.total_faces + 1, a.scan_partials, s);
  launch_exclusive_scan_u32(a.used, a.total_points + 1, a.scan_partials, s);
  hipLaunchKernelGGL(k_mb_values_used, grid_of(a.total_ap), kBlock, 0, s, a);
  launch_exclusive_scan_u32(a.vused, a.total_ap + 1, a.scan_partials, s);
  hipLaunchKernelGGL(k_mb_layout, 1, kBlock, 0, s, a);
  hipLaunchKernelGGL(k_mb_faces_out, grid_of(a.total_faces), kBlock, 0, s, a);
  hipLaunchKernelGGL(k_mb_attributes_out, grid_of(a.total_ap), kBlock, 0, s, a);
}
void launch_widen_indices(const MbWiden* items_dev, uint32_t n_items, uint32_t total, const uint8_t* src, uint32_t* dst, hipStream_t s) {
  if (n_items && total) hipLaunchKernelGGL(k_mb_widen, grid_of(total), kBlock, 0, s, items_dev, n_items, total, src, dst);
}

}  // namespace dmi
