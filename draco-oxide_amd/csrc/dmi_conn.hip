// dmi_conn.hip — the order-free half of the connectivity stage on the device (gfx950, wave64), ONE launch per kernel for all meshes
// of a batch (concatenated arrays with per-mesh offsets; ids stay mesh-local so that every mesh's slice is the table the host walks
// and the relabelling kernels consume as they are).
//
// Reference stages (paths relative to draco-oxide/src/):
//   k_conn_faces      core/corner_table/mod.rs:84-118     conn_faces = pos.p2v[face], the unused-vertex panic (:105-108) as a flag
//   k_conn_fill/match core/corner_table/mod.rs:252-340    compute_table: opposite corners.  The reference matches half-edges in corner
//                     order with per-vertex buckets; for a mesh without vertex-degenerate faces and with at most two faces on every
//                     undirected edge the result does not depend on that order (corner c is linked to the one corner carrying the
//                     reverse half-edge unless both have the same tip, Q22) — the argument of host_conn.cpp's order-free builder,
//                     which this restates with device atomics.  Meshes outside that class are FLAGGED, and the host runs the
//                     reference's serial walks on them (host_conn.cpp): nothing is approximated here.
//   k_conn_vertices   core/corner_table/mod.rs:342-416    left-most corners when every vertex has one fan (else flagged: the serial
//                     walk splits such vertices), and GenericCornerTable::is_on_boundary (:36-38) per vertex for the sequencer
// Attribute corner tables (attribute_corner_table.rs:16-137) stay with the host builder: one comparison per edge, no bucket matching.
// Then the coding-order relabelling of the meshes of a batch (the single-large-mesh form lives in dmi_relabel.hip):
//   k_rl_rank / k_rl_keys / k_rl_place / k_rl_sort_buckets / k_rl_remap / k_rl_seq — a counting sort by key, no library sort
#include "dmi_device.hpp"
#include <algorithm>

namespace dmi {
namespace {

constexpr int kBlock = 256;
constexpr uint32_t kNoneD = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t cnext(uint32_t c) { return (c % 3u == 2u) ? c - 2u : c + 1u; }
__device__ __forceinline__ uint32_t cprev(uint32_t c) { return (c % 3u == 0u) ? c + 2u : c - 1u; }
inline uint32_t grid_of(uint64_t n, uint32_t per_thread = 1) {
  const uint64_t g = (n + (uint64_t)kBlock * per_thread - 1) / ((uint64_t)kBlock * per_thread);
  return (uint32_t)(g > 65535ull * 32 ? 65535ull * 32 : (g ? g : 1));
}

// mesh of a global face / vertex index: the last mesh whose offset is ≤ x (offsets ascending; empty meshes share an offset with their successor)
template <class Get>
__device__ __forceinline__ uint32_t find_mesh(uint32_t M, uint32_t x, Get off) {
  uint32_t lo = 0, hi = M;   // invariant: off(lo) ≤ x < off(hi) with off(M) = ∞
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off(mid) <= x) lo = mid; else hi = mid; }
  return lo;
}

// ---- exclusive prefix sum over uint32 (in place): block sums → one block scans them → blocks rescan their tile ----
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t u = __shfl_up(v, off, 64); if ((int)(threadIdx.x & 63) >= off) v += u; }
  return v;
}
// exclusive scan of one value per thread across the block; returns the thread's offset, *total = block sum
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[kBlock / 64 + 1];
  const uint32_t inc = wave_inclusive(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();   // (wsum may still be read by a previous call)
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) { if (w < wave) base += wsum[w]; all += wsum[w]; }
  if (total) *total = all;
  return base + inc - v;
}
__global__ __launch_bounds__(kBlock) void k_scan_reduce(const uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ partials) {
  const uint32_t base = blockIdx.x * kScanTile;
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) { const uint32_t i = base + k * kBlock + threadIdx.x; if (i < n) sum += data[i]; }
  uint32_t total;
  (void)block_exclusive(sum, &total);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
__global__ __launch_bounds__(kBlock) void k_scan_partials(uint32_t* __restrict__ partials, uint32_t n_partials) {
  // one block: every thread owns a contiguous slice of the partials
  const uint32_t per = (n_partials + kBlock - 1) / kBlock;
  const uint32_t lo = min(n_partials, threadIdx.x * per), hi = min(n_partials, lo + per);
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += partials[i];
  uint32_t run = block_exclusive(sum, nullptr);
  for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = partials[i]; partials[i] = run; run += v; }
}
__global__ __launch_bounds__(kBlock) void k_scan_apply(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ partials) {
  // thread t owns items [t·kScanItems, (t+1)·kScanItems) of the tile (a serial scan in registers, then a block scan of the thread sums)
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  uint32_t v[kScanItems], sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) { v[k] = (base + k < n) ? data[base + k] : 0u; sum += v[k]; }
  uint32_t run = block_exclusive(sum, nullptr) + partials[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) { if (base + k < n) data[base + k] = run; run += v[k]; }
}

// A flag is raised by at most a few atomics however many threads find its cause (same-address atomics serialise at ≈ 11 ns each).
__device__ __forceinline__ void raise(uint32_t* flags, uint32_t m, uint32_t bit) {
  if (!(__atomic_load_n(&flags[m], __ATOMIC_RELAXED) & bit)) atomicOr(&flags[m], bit);
}
// per-mesh maximum: one atomic per wavefront when all its lanes are in the same mesh
__device__ __forceinline__ void mesh_max(uint32_t* dst, uint32_t m, uint32_t v, bool have) {
  const uint64_t mask = __ballot(have);
  if (mask == 0ull) return;   // (wave-uniform)
  const uint32_t m0 = __shfl(m, __ffsll((long long)mask) - 1, 64);
  if (__ballot(have && m != m0) == 0ull) {
    uint32_t x = have ? v : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = max(x, (uint32_t)__shfl_down(x, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&dst[m0], x);
  } else if (have) {
    atomicMax(&dst[m], v);
  }
}

// ---- universal corner tables of a batch ----
// per face: vertex ids (through the position map), range / degenerate checks, half-edge counts per bucket (= smaller endpoint of the
// edge), a corner per vertex, largest vertex id per mesh.
// The half-edge of corner j runs from v[j+1] to v[j+2]: its bucket is the smaller endpoint, and its place in the bucket — any numbering of a bucket's
// half-edges will do, k_conn_match looks at all of them — is parked in `opp` for k_conn_fill.  Round 6: a block takes a TILE of kBlock · FPT consecutive
// faces and counts the buckets within [the tile's smallest bucket, + kConnWindow) in LDS: a returning LDS atomic per half-edge, then ONE returning global
// atomic per bucket the tile touched (its count in, the tile's first place out) instead of one per half-edge — memory-side atomics were the kernel
// (10M faces: 640 µs, 120 without them; scripts/experiments/conn_faces_probe.hip) and a mesh whose vertex ids follow its face order has a tile's ≈ 3 · T
// half-edges in ≈ T buckets: 160 µs.  Buckets outside the window take the global atomic as before (a mesh with scattered ids: 2.4 ms either way).
// The corner kept per vertex goes through the window too (16-bit tile-relative id, one global store per vertex and tile).
constexpr int kConnWindow = 8192;
template <int FPT>
__global__ __launch_bounds__(kBlock) void k_conn_faces(const ConnArgs a) {
  __shared__ uint32_t cnt[kConnWindow];   // half-edges of the tile per bucket, then the tile's first place in the bucket
  __shared__ uint16_t fst[kConnWindow];   // some corner (tile-relative) of the vertex, 0xFFFF = none in this tile
  __shared__ uint32_t smin;
  static_assert(3 * kBlock * FPT < 65535 && kConnWindow % (kBlock * 8) == 0, "tile-relative corner ids are 16 bits");
  const uint32_t tile0 = blockIdx.x * (uint32_t)(kBlock * FPT);
  uint32_t gv[FPT][3], rank[FPT][3], foff[FPT];   // (gv: vert_off + vertex — the index into ecount / first)
  bool ok[FPT];
  uint32_t mymin = kNoneD, cur_m = kNoneD, cur_max = 0;   // largest vertex id seen for mesh cur_m (a thread's faces ascend, so do their meshes)
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    const uint32_t f = tile0 + j * kBlock + threadIdx.x;
    ok[j] = f < a.total_faces;
    if (!ok[j]) continue;
    const uint32_t m = a.M == 1 ? 0u : find_mesh(a.M, f, [&](uint32_t k) { return a.meshes[k].face_off; });
    if (m != cur_m) { if (cur_m != kNoneD) atomicMax(&a.vmax[cur_m], cur_max); cur_m = m; cur_max = 0; }
    const ConnMeshDesc d = a.meshes[m];
    foff[j] = d.face_off;
    uint32_t v[3];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t p = a.faces[3ull * f + k];
      if (p >= d.num_points) { bad = true; v[k] = 0; continue; }
      v[k] = d.p2v_off != kNoneD ? a.p2v[d.p2v_off + p] : p;
      if (v[k] >= d.Vcap) { bad = true; v[k] = 0; }
    }
    if (a.c2v != a.faces) { a.c2v[3ull * f] = v[0]; a.c2v[3ull * f + 1] = v[1]; a.c2v[3ull * f + 2] = v[2]; }
    if (bad) { raise(a.flags, m, CONN_BAD_INDEX); ok[j] = false; continue; }
    if (v[0] == v[1] || v[1] == v[2] || v[0] == v[2]) { raise(a.flags, m, CONN_DEGENERATE); ok[j] = false; continue; }
    cur_max = max(cur_max, max(v[0], max(v[1], v[2])));
#pragma unroll
    for (int k = 0; k < 3; ++k) gv[j][k] = d.vert_off + v[k];
    mymin = min(mymin, min(gv[j][0], min(gv[j][1], gv[j][2])));
  }
  mesh_max(a.vmax, cur_m, cur_max, cur_m != kNoneD);
  if (threadIdx.x == 0) smin = kNoneD;
  for (int i = threadIdx.x; i < kConnWindow; i += kBlock) { cnt[i] = 0; fst[i] = 0xFFFFu; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mymin = min(mymin, (uint32_t)__shfl_down(mymin, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && mymin != kNoneD) atomicMin(&smin, mymin);
  __syncthreads();
  const uint32_t base = smin;   // (kNoneD: no valid face in the tile — nothing below touches the window)
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    if (!ok[j]) continue;
    const uint32_t tf = j * kBlock + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t low = min(gv[j][k], gv[j][(k + 1) % 3]), dl = low - base;
      rank[j][k] = dl < (uint32_t)kConnWindow ? atomicAdd(&cnt[dl], 1u) : atomicAdd(&a.ecount[low], 1u);
      // SOME corner of the vertex (plain stores: whichever face writes last wins); k_conn_vertices finds the smallest one of the fan itself
      const uint32_t e = gv[j][k] - base;
      if (e < (uint32_t)kConnWindow) fst[e] = (uint16_t)(3u * tf + k);
      else a.first[gv[j][k]] = 3u * (tile0 + tf - foff[j]) + k;
    }
  }
  __syncthreads();
  if (base != kNoneD) {
    for (int it = 0; it < kConnWindow / kBlock; it += 8) {
      uint32_t c[8], r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = cnt[(it + u) * kBlock + threadIdx.x];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = c[u] ? atomicAdd(&a.ecount[base + (it + u) * kBlock + threadIdx.x], c[u]) : 0u;   // (eight in flight)
#pragma unroll
      for (int u = 0; u < 8; ++u) cnt[(it + u) * kBlock + threadIdx.x] = r[u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t x = fst[(it + u) * kBlock + threadIdx.x];
        if (x == 0xFFFFu) continue;
        const uint32_t g = base + (it + u) * kBlock + threadIdx.x;   // the corner's id is local to the mesh the vertex belongs to
        const uint32_t fo = a.M == 1 ? a.meshes[0].face_off : a.meshes[find_mesh(a.M, g, [&](uint32_t k) { return a.meshes[k].vert_off; })].face_off;
        a.first[g] = (uint32_t)(3ull * tile0 + x - 3ull * fo);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    if (!ok[j]) continue;
    const uint32_t f = tile0 + j * kBlock + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t dl = min(gv[j][k], gv[j][(k + 1) % 3]) - base;
      a.opp[3ull * f + (k + 2) % 3] = rank[j][k] + (dl < (uint32_t)kConnWindow ? cnt[dl] : 0u);
    }
  }
}
// per corner: its half-edge (source = vertex(next), sink = vertex(prev)) into the bucket of the smaller endpoint, keyed by the larger one
// and the direction (bit 31: the half-edge runs from the larger endpoint down)
__global__ __launch_bounds__(kBlock) void k_conn_fill(const ConnArgs a) {
  const uint64_t C = 3ull * a.total_faces;
  for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < C; c += (uint64_t)gridDim.x * kBlock) {
    const uint32_t f = (uint32_t)(c / 3u);
    const uint32_t m = find_mesh(a.M, f, [&](uint32_t k) { return a.meshes[k].face_off; });
    if (a.flags[m] & (CONN_BAD_INDEX | CONN_DEGENERATE)) continue;   // (the host builds this mesh's tables)
    const ConnMeshDesc d = a.meshes[m];
    const uint32_t lc = (uint32_t)(c - 3ull * d.face_off);
    const uint64_t cb = 3ull * d.face_off;
    const uint32_t src = a.c2v[cb + cnext(lc)], snk = a.c2v[cb + cprev(lc)];
    const uint32_t low = min(src, snk);
    const uint32_t slot = a.ecount[d.vert_off + low] + a.opp[c];   // ecount holds the bucket starts by now, opp the arrival number from k_conn_faces
    a.he_key[slot] = src < snk ? snk : (src | 0x80000000u);
    a.he_corner[slot] = lc;
  }
}
__global__ __launch_bounds__(kBlock) void k_conn_match(const ConnArgs a) {
  const uint64_t C = 3ull * a.total_faces;
  for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < C; c += (uint64_t)gridDim.x * kBlock) {
    const uint32_t f = (uint32_t)(c / 3u);
    const uint32_t m = find_mesh(a.M, f, [&](uint32_t k) { return a.meshes[k].face_off; });
    if (a.flags[m] & (CONN_BAD_INDEX | CONN_DEGENERATE)) { a.opp[c] = kNoneD; continue; }
    const ConnMeshDesc d = a.meshes[m];
    const uint32_t lc = (uint32_t)(c - 3ull * d.face_off);
    const uint64_t cb = 3ull * d.face_off;
    const uint32_t tip = a.c2v[c], src = a.c2v[cb + cnext(lc)], snk = a.c2v[cb + cprev(lc)];
    const uint32_t low = min(src, snk);
    const uint32_t mine = src < snk ? snk : (src | 0x80000000u), against = mine ^ 0x80000000u;
    uint32_t same = 0, rev = 0, found = kNoneD;
    for (uint32_t s = a.ecount[d.vert_off + low], e = a.ecount[d.vert_off + low + 1]; s < e; ++s) {
      const uint32_t k = a.he_key[s];
      same += k == mine;
      if (k == against) { ++rev; found = a.he_corner[s]; }
    }
    if (same + rev > 2) raise(a.flags, m, CONN_NONMANIFOLD_EDGE);
    const uint32_t o = (rev == 1 && same == 1 && a.c2v[cb + found] != tip) ? found : kNoneD;
    a.opp[c] = o;
    if (o == kNoneD) raise(a.flags, m, CONN_HAS_BOUNDARY);
  }
}
// per vertex: left-most corner of its (single) fan, the boundary flag; every corner of the fan is ticked off in cdone
__global__ __launch_bounds__(kBlock) void k_conn_vertices(const ConnArgs a) {
  for (uint32_t gv = blockIdx.x * kBlock + threadIdx.x; gv < a.total_verts; gv += gridDim.x * kBlock) {
    const uint32_t m = find_mesh(a.M, gv, [&](uint32_t k) { return a.meshes[k].vert_off; });
    const ConnMeshDesc d = a.meshes[m];
    const uint32_t v = gv - d.vert_off;
    a.lmc[gv] = kNoneD;
    a.on_boundary[gv] = 0;
    if (a.flags[m] & (CONN_BAD_INDEX | CONN_DEGENERATE | CONN_NONMANIFOLD_EDGE)) continue;
    if (d.F == 0 || v > a.vmax[m]) continue;   // ids past the largest referenced one are not vertices of the table (V = max + 1)
    const uint32_t c = a.first[gv];
    if (c == kNoneD) { raise(a.flags, m, CONN_UNUSED_VERTEX); continue; }
    const uint64_t cb = 3ull * d.face_off;
    const uint32_t* __restrict__ opp = a.opp + cb;
    uint8_t* __restrict__ done = a.cdone + cb;
    auto swing_left = [&](uint32_t x) { const uint32_t o = opp[cnext(x)]; return o == kNoneD ? kNoneD : cnext(o); };
    auto swing_right = [&](uint32_t x) { const uint32_t o = opp[cprev(x)]; return o == kNoneD ? kNoneD : cprev(o); };
    // The reference walks the corners in order and swings left from the FIRST (smallest) corner of the vertex (mod.rs:342-416): an open fan ends at
    // its left-most corner wherever the walk started; a closed fan's "left-most" corner is the last one met before the walk is back at that smallest
    // corner, i.e. swing_right(smallest).  c is just some corner of the vertex (round 5: k_conn_faces no longer pays three atomicMin per face to
    // know the smallest — the walk visits every corner of a closed fan anyway and keeps the minimum).
    uint32_t left = c, x = swing_left(c), steps = 0, smallest = c;
    done[c] = 1;
    while (x != kNoneD && x != c && steps < (1u << 20)) { done[x] = 1; left = x; smallest = min(smallest, x); x = swing_left(x); ++steps; }
    if (steps >= (1u << 20)) { raise(a.flags, m, CONN_MULTI_FAN); continue; }
    if (x == kNoneD) {   // open fan: the corners to the right of c
      steps = 0;
      for (uint32_t r = swing_right(c); r != kNoneD && steps < (1u << 20); r = swing_right(r), ++steps) done[r] = 1;
    } else {
      left = swing_right(smallest);   // (closed: never none)
    }
    a.lmc[gv] = left;
    a.on_boundary[gv] = opp[cnext(left)] == kNoneD ? 1 : 0;
  }
}
// a corner no fan ticked off: its vertex has several fans (the serial walk splits it into several vertices: mod.rs:368-385)
__global__ __launch_bounds__(kBlock) void k_conn_check(const ConnArgs a) {
  const uint64_t C = 3ull * a.total_faces;
  for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < C; c += (uint64_t)gridDim.x * kBlock) {
    if (a.cdone[c]) continue;
    const uint32_t m = find_mesh(a.M, (uint32_t)(c / 3u), [&](uint32_t k) { return a.meshes[k].face_off; });
    if (a.flags[m] & (CONN_BAD_INDEX | CONN_DEGENERATE | CONN_NONMANIFOLD_EDGE)) continue;
    raise(a.flags, m, CONN_MULTI_FAN);
  }
}

// ---- attribute corner tables of a batch (attribute_corner_table.rs:16-137) ----
// The host builder (host_conn.cpp build_attribute_into / finish_attribute) restated per corner / per vertex: every step is order-free.
constexpr uint32_t kConnSkip = CONN_BAD_INDEX | CONN_DEGENERATE | CONN_NONMANIFOLD_EDGE | CONN_MULTI_FAN | CONN_UNUSED_VERTEX;
__device__ __forceinline__ uint32_t att_item_of_corner(const AttArgs& t, uint32_t g) { return find_mesh(t.n_items, g, [&](uint32_t k) { return t.items[k].corner_off; }); }
__device__ __forceinline__ uint32_t att_item_of_vertex(const AttArgs& t, uint32_t g) { return find_mesh(t.n_items, g, [&](uint32_t k) { return t.items[k].vert_off; }); }
// seam test per edge (:44-63): a boundary, or the attribute's values differ at either shared endpoint
__global__ __launch_bounds__(kBlock) void k_att_seams(const ConnArgs a, const AttArgs t) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < t.total_corners; g += gridDim.x * kBlock) {
    const uint32_t i = att_item_of_corner(t, g);
    const AttItemDesc it = t.items[i];
    const ConnMeshDesc d = a.meshes[it.mesh];
    if (a.flags[it.mesh] & kConnSkip) continue;
    const uint32_t c = g - it.corner_off;
    if (c >= 3u * d.F) continue;
    const uint64_t cb = 3ull * d.face_off;
    const uint32_t o = a.opp[cb + c];
    const uint32_t* __restrict__ c2v = a.c2v + cb;
    uint8_t* __restrict__ vs = t.vseam + it.vert_off;
    if (o == kNoneD) { t.seam[g] = 1; vs[c2v[cnext(c)]] = 1; vs[c2v[cprev(c)]] = 1; continue; }
    if (o < c) continue;
    const uint32_t* __restrict__ f = a.faces + cb;
    const uint32_t pa = f[cnext(c)], pb = f[cprev(o)], pc = f[cprev(c)], pd = f[cnext(o)];
    auto val = [&](uint32_t p) { return it.map_off != kNoneD ? a.p2v[it.map_off + p] : p; };
    if ((pa != pb && val(pa) != val(pb)) || (pc != pd && val(pc) != val(pd))) {
      t.seam[g] = 1; t.seam[it.corner_off + o] = 1;
      vs[c2v[cnext(c)]] = 1; vs[c2v[cprev(c)]] = 1; vs[c2v[cnext(o)]] = 1; vs[c2v[cprev(o)]] = 1;
      if (!__atomic_load_n(&t.info[i].interior, __ATOMIC_RELAXED)) atomicOr(&t.info[i].interior, 1u);
    }
  }
}
// seam-aware fan start of vertex v (:101-113) and the walk to the right over the UNIVERSAL table (:116-133)
struct AttFan {
  const uint32_t* __restrict__ opp; const uint8_t* __restrict__ seam; uint32_t lmc_v;
  __device__ __forceinline__ uint32_t a_swing_left(uint32_t c) const { const uint32_t n = cnext(c); if (seam[n]) return kNoneD; const uint32_t o = opp[n]; return o == kNoneD ? kNoneD : cnext(o); }
  __device__ __forceinline__ uint32_t u_swing_right(uint32_t c) const { const uint32_t o = opp[cprev(c)]; return o == kNoneD ? kNoneD : cprev(o); }
  __device__ __forceinline__ uint32_t start() const { uint32_t first = lmc_v, steps = 0; for (uint32_t n; (n = a_swing_left(first)) != kNoneD && n != lmc_v && steps < (1u << 20); ++steps) first = n; return first; }
};
__global__ __launch_bounds__(kBlock) void k_att_count(const ConnArgs a, const AttArgs t) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < t.total_verts; g += gridDim.x * kBlock) {
    const AttItemDesc it = t.items[att_item_of_vertex(t, g)];
    const ConnMeshDesc d = a.meshes[it.mesh];
    const uint32_t v = g - it.vert_off;
    if ((a.flags[it.mesh] & kConnSkip) || d.F == 0 || v > a.vmax[it.mesh]) continue;   // (count stays 0)
    uint32_t k = 1;
    if (t.vseam[g]) {
      const AttFan fan{a.opp + 3ull * d.face_off, t.seam + it.corner_off, a.lmc[d.vert_off + v]};
      const uint32_t first = fan.start();
      uint32_t steps = 0;
      for (uint32_t cur = fan.u_swing_right(first); cur != kNoneD && cur != first && steps < (1u << 20); cur = fan.u_swing_right(cur), ++steps) k += fan.seam[cnext(cur)];
    }
    t.count[g] = k;
  }
}
// corners of vertices no seam touches: one attribute vertex, the universal left-most corner; every corner's opposite
__global__ __launch_bounds__(kBlock) void k_att_corners(const ConnArgs a, const AttArgs t) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < t.total_corners; g += gridDim.x * kBlock) {
    const AttItemDesc it = t.items[att_item_of_corner(t, g)];
    const ConnMeshDesc d = a.meshes[it.mesh];
    if (a.flags[it.mesh] & kConnSkip) continue;
    const uint32_t c = g - it.corner_off;
    if (c >= 3u * d.F) continue;
    const uint64_t cb = 3ull * d.face_off;
    t.opp[g] = t.seam[g] ? kNoneD : a.opp[cb + c];
    const uint32_t v = a.c2v[cb + c];
    if (!t.vseam[it.vert_off + v]) t.c2v[g] = t.count[it.vert_off + v] - t.count[it.vert_off];
  }
}
// per vertex: left-most corners; the vertices ON a seam walk their fan and hand out their attribute vertices
__global__ __launch_bounds__(kBlock) void k_att_vertices(const ConnArgs a, const AttArgs t) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < t.total_verts; g += gridDim.x * kBlock) {
    const uint32_t i = att_item_of_vertex(t, g);
    const AttItemDesc it = t.items[i];
    const ConnMeshDesc d = a.meshes[it.mesh];
    const uint32_t v = g - it.vert_off;
    if ((a.flags[it.mesh] & kConnSkip) || d.F == 0 || v > a.vmax[it.mesh]) continue;
    const uint32_t base0 = t.count[it.vert_off];
    uint32_t id = t.count[g] - base0;
    uint32_t* __restrict__ lmc = t.lmc + base0;   // (the scan is global: the items' attribute vertices lie one after the other)
    if (v == 0) { t.info[i].num_vertices = t.count[it.vert_off + a.vmax[it.mesh] + 1] - base0; t.info[i].pad = base0; t.info[i].done = 1; }   // pad = where the item's left-most corners start
    if (!t.vseam[g]) { lmc[id] = a.lmc[d.vert_off + v]; continue; }
    const AttFan fan{a.opp + 3ull * d.face_off, t.seam + it.corner_off, a.lmc[d.vert_off + v]};
    const uint32_t first = fan.start();
    uint32_t* __restrict__ c2v = t.c2v + it.corner_off;
    c2v[first] = id;
    lmc[id] = first;
    uint32_t steps = 0;
    for (uint32_t cur = fan.u_swing_right(first); cur != kNoneD && cur != first && steps < (1u << 20); cur = fan.u_swing_right(cur), ++steps) {
      if (fan.seam[cnext(cur)]) { ++id; lmc[id] = cur; }
      c2v[cur] = id;
    }
  }
}

// ---- coding-order relabelling of a batch of (small) meshes: dmi_relabel.hip's steps with per-mesh descriptors, one launch per step ----
// (reference seam and the argument why this is a pure relabelling: dmi_relabel.hip / DESIGN.md §3)
template <class F> __device__ __forceinline__ uint32_t find_item(const RelabelItem* __restrict__ items, uint32_t n, uint32_t x, F off) {
  return find_mesh(n, x, [&](uint32_t k) { return off(items[k]); });
}
// rank[vertex(seq[k])] = k
__global__ __launch_bounds__(kBlock) void k_rl_rank(const RelabelBatch b) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < b.total_seq; g += gridDim.x * kBlock) {
    const RelabelItem& it = b.items[find_item(b.items, b.n_items, g, [](const RelabelItem& x) { return x.seq_off; })];
    const uint32_t k = g - it.seq_off;
    b.rank[it.vert_off + it.c2v[it.seq[k]]] = k;
  }
}
// key[f] = smallest sequence index among the face's vertices (n_seq when none was coded); bucket sizes per key
__global__ __launch_bounds__(kBlock) void k_rl_keys(const RelabelBatch b) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < b.total_faces; g += gridDim.x * kBlock) {
    const RelabelItem& it = b.items[find_item(b.items, b.n_items, g, [](const RelabelItem& x) { return x.face_off; })];
    const uint32_t f = g - it.face_off;
    if (it.plain) { if (f == 0) b.count[it.key_off] = it.F; continue; }   // (no order of its own, but its faces keep their places in the batch's position space: the scan below runs over all items)
    const uint32_t* r = b.rank + it.vert_off;
    const uint32_t m = min(r[it.c2v[3ull * f]], min(r[it.c2v[3ull * f + 1]], r[it.c2v[3ull * f + 2]]));
    const uint32_t key = m == kNoneD ? it.n_seq : m;
    b.key[g] = key;
    b.new_face[g] = atomicAdd(&b.count[it.key_off + key], 1u);   // arrival number in the bucket (parked in new_face until k_rl_sort_buckets writes it)
  }
}
// faces into their buckets (the scan of the bucket sizes is a global face position: the keys of mesh m occupy [key_off[m], key_off[m+1]))
__global__ __launch_bounds__(kBlock) void k_rl_place(const RelabelBatch b) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < b.total_faces; g += gridDim.x * kBlock) {
    const RelabelItem& it = b.items[find_item(b.items, b.n_items, g, [](const RelabelItem& x) { return x.face_off; })];
    if (it.plain) continue;
    b.order[b.count[it.key_off + b.key[g]] + b.new_face[g]] = g - it.face_off;
  }
}
// faces of equal key in face order (the host form's stable counting sort): a bucket holds the faces around one vertex — a handful
__global__ __launch_bounds__(kBlock) void k_rl_sort_buckets(const RelabelBatch b) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < b.total_keys; g += gridDim.x * kBlock) {
    const uint32_t lo = b.count[g], hi = b.count[g + 1];
    if (hi <= lo) continue;
    if (b.items[find_item(b.items, b.n_items, g, [](const RelabelItem& x) { return x.key_off; })].plain) continue;
    uint32_t* o = b.order;
    for (uint32_t i = lo + 1; i < hi; ++i) {   // insertion sort
      const uint32_t v = o[i];
      uint32_t j = i;
      while (j > lo && o[j - 1] > v) { o[j] = o[j - 1]; --j; }
      o[j] = v;
    }
    const RelabelItem& it = b.items[find_item(b.items, b.n_items, g, [](const RelabelItem& x) { return x.key_off; })];
    for (uint32_t i = lo; i < hi; ++i) b.new_face[it.face_off + o[i]] = i - it.face_off;
  }
}
__device__ __forceinline__ uint32_t map_corner(uint32_t c, const uint32_t* __restrict__ new_face) { return c == kNoneD ? kNoneD : 3u * new_face[c / 3u] + c % 3u; }
// one thread per NEW corner of every table: c2r_out[c2] = rank[c2v[c]], opp_out[c2] = map(opp[c]) with c = 3·order[c2 / 3] + c2 % 3 — order and
// new_face are the universal table's of the same mesh (order_item), the ranks the table's own
__global__ __launch_bounds__(kBlock) void k_rl_remap(const RelabelBatch b) {
  const uint64_t C = 3ull * b.total_remap_faces;
  for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < C; g += (uint64_t)gridDim.x * kBlock) {
    const uint32_t gf = (uint32_t)(g / 3u), k = (uint32_t)(g - 3ull * gf);
    const RelabelItem& it = b.items[find_item(b.items, b.n_items, gf, [](const RelabelItem& x) { return x.remap_off; })];
    const uint32_t uf = b.items[it.order_item].face_off;
    const uint32_t f2 = gf - it.remap_off, c2 = 3u * f2 + k;
    if (it.plain) { it.c2r[c2] = b.rank[it.vert_off + it.c2v[c2]]; it.opp_out[c2] = it.opp[c2]; continue; }   // the mesh's own face order
    const uint32_t c = 3u * b.order[uf + f2] + k;
    it.c2r[c2] = b.rank[it.vert_off + it.c2v[c]];
    it.opp_out[c2] = map_corner(it.opp[c], b.new_face + uf);
  }
}
__global__ __launch_bounds__(kBlock) void k_rl_seq(const RelabelBatch b) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < b.total_seq; g += gridDim.x * kBlock) {
    const RelabelItem& it = b.items[find_item(b.items, b.n_items, g, [](const RelabelItem& x) { return x.seq_off; })];
    const uint32_t k = g - it.seq_off, c = it.seq[k];
    it.seq_out[k] = it.plain ? c : map_corner(c, b.new_face + b.items[it.order_item].face_off);
    it.s2p[k] = it.c2p[c];
  }
}
// s2v[k] = point_to_value[s2p[k]] for every attribute of the batch that carries a map
__global__ __launch_bounds__(kBlock) void k_compose_batch(const ComposeItem* __restrict__ items, uint32_t n_items, uint32_t total) {
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < total; g += gridDim.x * kBlock) {
    const ComposeItem& it = items[find_mesh(n_items, g, [&](uint32_t k) { return items[k].off; })];
    const uint32_t k = g - it.off;
    it.s2v[k] = it.p2v[it.s2p[k]];   // (maps and faces were range-checked by the connectivity stage)
  }
}

}  // namespace

void launch_exclusive_scan_u32(uint32_t* data, uint32_t n, uint32_t* partials, hipStream_t s) {
  if (!n) return;
  const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_scan_reduce, tiles, kBlock, 0, s, data, n, partials);
  hipLaunchKernelGGL(k_scan_partials, 1, kBlock, 0, s, partials, tiles);
  hipLaunchKernelGGL(k_scan_apply, tiles, kBlock, 0, s, data, n, partials);
}
size_t scan_partials_words(uint32_t n) { return (size_t)(n + kScanTile - 1) / kScanTile + 1; }

// flags / vmax / cdone zeroed, first filled with DMI_NONE, ecount zeroed by the caller's memsets (conn_tables_clear); efill is unused since round 4
// (the arrival numbers of the counting atomics replace the second round of atomics)
// The host's copy of the opposite corners with ids 4·face + k (host_conn.cpp Enc4: the serial walks then find face and position by shift and mask).
// One streaming pass, beside a read-back that is bound by the link anyway.
__global__ __launch_bounds__(kBlock) void k_opp_quad(const uint32_t* __restrict__ opp, uint64_t C, uint32_t* __restrict__ out) {
  for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < C; c += (uint64_t)gridDim.x * kBlock) {
    const uint32_t o = opp[c];
    out[c] = o == kNoneD ? kNoneD : o + o / 3u;
  }
}
void launch_opp_quad(const uint32_t* opp, uint64_t C, uint32_t* out, hipStream_t s) {
  if (C) hipLaunchKernelGGL(k_opp_quad, (uint32_t)std::min<uint64_t>((C + kBlock - 1) / kBlock, 8192), kBlock, 0, s, opp, C, out);
}

hipError_t conn_tables_clear(const ConnArgs& a, hipStream_t s) {
  const size_t nv = (size_t)a.total_verts + 1, C = 3ull * a.total_faces;
  ClearRanges r{};   // (one launch: five hipMemsetAsync before)
  r.add(a.flags, (size_t)a.M * 4); r.add(a.vmax, (size_t)a.M * 4); r.add(a.ecount, nv * 4); r.add(a.first, nv * 4, 0xFF); r.add(a.cdone, C);
  launch_clear_ranges(r, s);
  return hipGetLastError();
}
void launch_conn_tables(const ConnArgs& a, hipStream_t s) {
  if (!a.total_faces || !a.M) return;
  const uint64_t C = 3ull * a.total_faces;
  // tile size by the launch's size: large tiles touch fewer buckets per half-edge, small launches need the blocks
  if (a.total_faces >= (2u << 20)) hipLaunchKernelGGL(k_conn_faces<16>, grid_of(a.total_faces, 16), kBlock, 0, s, a);
  else if (a.total_faces >= (1u << 18)) hipLaunchKernelGGL(k_conn_faces<8>, grid_of(a.total_faces, 8), kBlock, 0, s, a);
  else hipLaunchKernelGGL(k_conn_faces<4>, grid_of(a.total_faces, 4), kBlock, 0, s, a);
  launch_exclusive_scan_u32(a.ecount, a.total_verts + 1, a.scan_partials, s);
  hipLaunchKernelGGL(k_conn_fill, grid_of(C), kBlock, 0, s, a);
  hipLaunchKernelGGL(k_conn_match, grid_of(C), kBlock, 0, s, a);
  hipLaunchKernelGGL(k_conn_vertices, grid_of(a.total_verts), kBlock, 0, s, a);
  hipLaunchKernelGGL(k_conn_check, grid_of(C), kBlock, 0, s, a);
}
hipError_t att_tables_clear(const AttArgs& t, hipStream_t s) {
  if (!t.n_items) return hipSuccess;
  ClearRanges r{};
  r.add(t.seam, t.total_corners); r.add(t.vseam, t.total_verts); r.add(t.count, ((size_t)t.total_verts + 1) * 4); r.add(t.info, (size_t)t.n_items * sizeof(AttInfo));
  launch_clear_ranges(r, s);
  return hipGetLastError();
}
void launch_att_tables(const ConnArgs& a, const AttArgs& t, hipStream_t s) {
  if (!t.n_items || !t.total_corners) return;
  hipLaunchKernelGGL(k_att_seams, grid_of(t.total_corners), kBlock, 0, s, a, t);
  hipLaunchKernelGGL(k_att_count, grid_of(t.total_verts), kBlock, 0, s, a, t);
  launch_exclusive_scan_u32(t.count, t.total_verts + 1, t.scan_partials, s);
  hipLaunchKernelGGL(k_att_corners, grid_of(t.total_corners), kBlock, 0, s, a, t);
  hipLaunchKernelGGL(k_att_vertices, grid_of(t.total_verts), kBlock, 0, s, a, t);
}
void launch_relabel_batch(const RelabelBatch& b, hipStream_t s) {
  if (!b.n_items || !b.total_faces) return;
  hipLaunchKernelGGL(k_rl_rank, grid_of(b.total_seq), kBlock, 0, s, b);
  if (b.any_sorted) {
    hipLaunchKernelGGL(k_rl_keys, grid_of(b.total_faces), kBlock, 0, s, b);
    launch_exclusive_scan_u32(b.count, b.total_keys + 1, b.scan_partials, s);
    hipLaunchKernelGGL(k_rl_place, grid_of(b.total_faces), kBlock, 0, s, b);
    hipLaunchKernelGGL(k_rl_sort_buckets, grid_of(b.total_keys), kBlock, 0, s, b);
  }
  hipLaunchKernelGGL(k_rl_remap, grid_of(3ull * b.total_remap_faces), kBlock, 0, s, b);
  hipLaunchKernelGGL(k_rl_seq, grid_of(b.total_seq), kBlock, 0, s, b);
}
void launch_compose_batch(const ComposeItem* items_dev, uint32_t n_items, uint32_t total, hipStream_t s) {
  if (n_items && total) hipLaunchKernelGGL(k_compose_batch, grid_of(total), kBlock, 0, s, items_dev, n_items, total);
}

}  // namespace dmi
