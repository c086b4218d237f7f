// dmi_relabel.hip — coding-order relabelling of the connectivity inputs on the device (job creation of large meshes).
//
// The corner tables arrive in the mesh's own face / vertex numbering; every predictor walks them in the coding (Edgebreaker) order.
// dmi_job_create therefore re-indexes them once (a pure relabelling: the bitstream does not depend on internal corner / vertex ids):
//   vertices → their sequence index          rank[vertex(seq[k])] = k                          k_rank_scatter
//   faces    → ordered by the smallest sequence index among their (universal) vertices,        k_face_keys + a counting sort by key: bucket sizes,
//              faces of equal key in face order (= the host form's counting sort)              prefix sum, k_place_faces, k_sort_buckets (→ new_face)
//   corner-indexed arrays and the corner ids stored in `seq` / `opp` follow the new face order k_remap_table, k_remap_seq
//   seq → corner_to_point → point_to_value is composed once                                    k_remap_seq (s2p), k_compose_s2v
// ≈ 1 GB of streaming / scattered traffic for a 10M-triangle mesh: well under a millisecond of kernels, against 130 ms on 16 host
// threads (round 1) — what remains of job creation is the PCIe upload of the caller's tables.  The host form (dmi_job.cpp) is kept for
// small meshes, whose creation is bound by per-launch costs; both produce identical arrays (tests compare the encodes).
// Reference seam: the arrays are the flat view of ConnectivityEncoderOutput::Edgebreaker{corner_table, corners_of_edgebreaker} handed to
// attribute::encode_attributes (encode/attribute/mod.rs:13-93); sequence = Traverser::compute_seqeunce (shared/attribute/sequence.rs:48).
#include "dmi_device.hpp"

namespace dmi {
namespace {

constexpr int kBlock = 256;
constexpr uint32_t kNoneD = 0xFFFFFFFFu;

inline uint32_t grid_of(uint64_t n) { const uint64_t g = (n + kBlock - 1) / kBlock; return (uint32_t)(g > 65535 * 16 ? 65535 * 16 : (g ? g : 1)); }

__global__ __launch_bounds__(kBlock) void k_fill_u32(uint32_t* __restrict__ p, uint64_t n, uint32_t v) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) p[i] = v;
}

// rank[vertex(seq[k])] = k   (every vertex is emitted once: sequence.rs:41-46 — no two k write the same slot)
__global__ __launch_bounds__(kBlock) void k_rank_scatter(const uint32_t* __restrict__ seq, uint32_t n_seq, const uint32_t* __restrict__ c2v, uint32_t* __restrict__ rank) {
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_seq; k += gridDim.x * kBlock) rank[c2v[seq[k]]] = k;
}

// key[f] = smallest sequence index among the face's vertices (none_key when no vertex of the face was coded); count[key]++
// (a bucket holds the faces around one coded vertex — a handful — so the atomics spread over as many addresses as there are vertices)
__global__ __launch_bounds__(kBlock) void k_face_keys(const uint32_t* __restrict__ c2v, const uint32_t* __restrict__ rank, uint32_t F, uint32_t none_key,
                                                      uint32_t* __restrict__ key, uint32_t* __restrict__ count, uint32_t* __restrict__ slot) {
  for (uint32_t f = blockIdx.x * kBlock + threadIdx.x; f < F; f += gridDim.x * kBlock) {
    const uint32_t a = rank[c2v[3 * (size_t)f]], b = rank[c2v[3 * (size_t)f + 1]], c = rank[c2v[3 * (size_t)f + 2]];
    const uint32_t m = min(a, min(b, c));
    const uint32_t k = (m == kNoneD) ? none_key : m;
    key[f] = k;
    slot[f] = atomicAdd(&count[k], 1u);   // the face's arrival number in its bucket: the placement below needs no second round of atomics
  }
}
// faces into their buckets (start = exclusive prefix sum of the bucket sizes), in the order the counting atomics handed out
__global__ __launch_bounds__(kBlock) void k_place_faces(const uint32_t* __restrict__ key, uint32_t F, const uint32_t* __restrict__ start, const uint32_t* __restrict__ slot, uint32_t* __restrict__ order) {
  for (uint32_t f = blockIdx.x * kBlock + threadIdx.x; f < F; f += gridDim.x * kBlock) order[start[key[f]] + slot[f]] = f;
}
// faces of equal key in face order (the host form's stable counting sort), then new_face[order[j]] = j
__global__ __launch_bounds__(kBlock) void k_sort_buckets(const uint32_t* __restrict__ start, uint32_t n_keys, uint32_t* __restrict__ order, uint32_t* __restrict__ new_face) {
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_keys; k += gridDim.x * kBlock) {
    const uint32_t lo = start[k], hi = start[k + 1];
    for (uint32_t i = lo + 1; i < hi; ++i) {   // insertion sort
      const uint32_t v = order[i];
      uint32_t j = i;
      while (j > lo && order[j - 1] > v) { order[j] = order[j - 1]; --j; }
      order[j] = v;
    }
    for (uint32_t i = lo; i < hi; ++i) new_face[order[i]] = i;
  }
}

__device__ __forceinline__ uint32_t map_corner(uint32_t c, const uint32_t* __restrict__ new_face) { return c == kNoneD ? kNoneD : 3u * new_face[c / 3u] + c % 3u; }

// One thread per NEW corner (coalesced writes of both outputs; the reads gather through `order`):
//   c2r_out[c2] = rank[c2v[c]],  opp_out[c2] = map(opp[c])      with c = 3·order[c2 / 3] + c2 % 3
__global__ __launch_bounds__(kBlock) void k_remap_table(const uint32_t* __restrict__ c2v, const uint32_t* __restrict__ opp, const uint32_t* __restrict__ rank,
                                                        const uint32_t* __restrict__ order, const uint32_t* __restrict__ new_face, uint64_t C,
                                                        uint32_t* __restrict__ c2r_out, uint32_t* __restrict__ opp_out) {
  for (uint64_t c2 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c2 < C; c2 += (uint64_t)gridDim.x * kBlock) {
    const uint32_t f2 = (uint32_t)(c2 / 3u), k = (uint32_t)(c2 - 3ull * f2);
    const uint64_t c = 3ull * order[f2] + k;
    c2r_out[c2] = rank[c2v[c]];
    opp_out[c2] = map_corner(opp[c], new_face);
  }
}

// seq_out[k] = map(seq[k]);  s2p_out[k] = corner_to_point[seq[k]]   (attribute_encoder.rs:332-338 reads attribute.get(point_idx(c)))
__global__ __launch_bounds__(kBlock) void k_remap_seq(const uint32_t* __restrict__ seq, uint32_t n_seq, const uint32_t* __restrict__ new_face, const uint32_t* __restrict__ c2p,
                                                      uint32_t* __restrict__ seq_out, uint32_t* __restrict__ s2p_out) {
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_seq; k += gridDim.x * kBlock) {
    const uint32_t c = seq[k];
    seq_out[k] = map_corner(c, new_face);
    s2p_out[k] = c2p[c];
  }
}

// The tables of a one-shot job whose attributes all ride the fused sweep stay in the mesh's OWN face order (round 5): the sweep reads fan rows, and
// the rows are built from (seq, c2r, opp) in whatever numbering these three agree on — only the ranks are needed:
//   c2r[c] = rank[c2v[c]] for every corner (streaming over c2v, gathers into the rank array);  s2p[k] = corner_to_point[seq[k]]
__global__ __launch_bounds__(kBlock) void k_corner_ranks(const uint32_t* __restrict__ c2v, const uint32_t* __restrict__ rank, uint64_t C, uint32_t* __restrict__ c2r) {
  for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < C; c += (uint64_t)gridDim.x * kBlock) c2r[c] = rank[c2v[c]];
}
// Face records (round 6; one-shot jobs in the mesh's own face order): frec[8f .. 8f+2] = the ranks of face f's three vertices, frec[8f+4 .. 8f+6] = its three
// opposite corners — 32 bytes per face, written once, so that a swing of the fan-row walk (k_build_fans_rec: the rank of the vertex it lands on AND the
// opposite corner it leaves by belong to one face) is ONE 32-byte read instead of a read of `c2r` and one of `opp` on different lines.
__global__ __launch_bounds__(kBlock) void k_face_records(const uint32_t* __restrict__ c2v, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ opp, uint32_t F,
                                                         uint4* __restrict__ frec) {
  for (uint32_t f = blockIdx.x * kBlock + threadIdx.x; f < F; f += gridDim.x * kBlock) {
    const size_t c = (size_t)f * 3;
    const uint32_t v0 = c2v[c], v1 = c2v[c + 1], v2 = c2v[c + 2];
    const uint32_t o0 = opp[c], o1 = opp[c + 1], o2 = opp[c + 2];
    frec[(size_t)f * 2] = make_uint4(rank[v0], rank[v1], rank[v2], 0u);
    frec[(size_t)f * 2 + 1] = make_uint4(o0, o1, o2, 0u);
  }
}
// for a table in the mesh's own face order (one read of the sequence, the two table entries of a corner next to each other in time):
// rank[c2v[seq[k]]] = k and s2p[k] = c2p[seq[k]]
__global__ __launch_bounds__(kBlock) void k_rank_and_points(const uint32_t* __restrict__ seq, uint32_t n_seq, const uint32_t* __restrict__ c2v, const uint32_t* __restrict__ c2p,
                                                            uint32_t* __restrict__ rank, uint32_t* __restrict__ s2p) {
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_seq; k += gridDim.x * kBlock) {
    const uint32_t c = seq[k];
    const uint32_t v = c2v[c];
    s2p[k] = c2p == c2v ? v : c2p[c];
    rank[v] = k;
  }
}

// s2v[k] = point_to_value[s2p[k]];  *bad |= 1 when a value index is out of range (reported by the host as an error code)
__global__ __launch_bounds__(kBlock) void k_compose_s2v(const uint32_t* __restrict__ s2p, uint32_t n_seq, const uint32_t* __restrict__ p2v, uint32_t num_points, uint32_t num_unique,
                                                        uint32_t* __restrict__ s2v, uint32_t* __restrict__ bad) {
  bool any = false;
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_seq; k += gridDim.x * kBlock) {
    const uint32_t p = s2p[k];
    uint32_t v = 0;
    if (p < num_points) v = p2v[p]; else any = true;
    if (v >= num_unique) { any = true; v = 0; }
    s2v[k] = v;
  }
  if (__syncthreads_or(any ? 1 : 0) && threadIdx.x == 0) atomicOr(bad, 1u);
}

// max over a u32 array (largest point index the faces reference), one atomic per block
__global__ __launch_bounds__(kBlock) void k_max_u32(const uint32_t* __restrict__ a, uint64_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) m = max(m, a[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down(m, off, 64));
  __shared__ uint32_t red[kBlock / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int w = 1; w < kBlock / 64; ++w) m = max(m, red[w]); atomicMax(out, m); }
}

// ---- the tile-sorted form of the quantize gather: inside tiles of 2^tile_log2 consecutive sequence entries the slots are ordered by
// point index, so that the lanes of a wavefront of k_seq_quantize read neighbouring points; slot j reads point out_p[j] and writes sequence
// entry out_dest[j].  A bitonic network over (point << 32 | position in the tile).  Tiles of up to 2^local_log2 entries (16 K: 128 KB of LDS)
// are one workgroup's: every stage in LDS.  Larger tiles (a 100M-triangle mesh wants 64–128 K entries: more than one ring of its coding
// order) run the strides below 2^local_log2 in LDS, block by block, and the longer ones as passes over a key array in global memory:
//   k_tile_sort_local(first)  →  for size = 2^(local+1) … 2^tile: { k_tile_merge_global per stride ≥ 2^local;  k_tile_sort_local(size) }
// The slots past n carry the largest key, sort to the end of their tile and are not written.  Job creation only.
constexpr uint32_t kSortThreads = 1024;
// One block of 2^local_log2 entries.  first: keys are formed from s2p and the stages size = 2 … 2^local_log2 run; otherwise the keys come from
// `keys` and only the strides below the block size of stage `size` run.  last: the results are written (out_p / out_dest), else the keys.
__global__ __launch_bounds__(kSortThreads) void k_tile_sort_local(const uint32_t* __restrict__ s2p, uint32_t n, uint32_t local_log2, uint32_t tile_log2, uint32_t size,
                                                                 int first, int last, uint64_t* __restrict__ keys, uint32_t* __restrict__ out_p, uint32_t* __restrict__ out_dest) {
  extern __shared__ uint64_t tile_keys[];
  const uint32_t T = 1u << local_log2, a0 = blockIdx.x << local_log2, tile_mask = (1u << tile_log2) - 1u;
  for (uint32_t k = threadIdx.x; k < T; k += kSortThreads) {
    const uint32_t i = a0 + k;
    tile_keys[k] = first ? (i < n ? (((uint64_t)s2p[i] << 32) | (i & tile_mask)) : ~0ull) : keys[i];
  }
  __syncthreads();
  const uint32_t size_lo = first ? 2u : size, size_hi = first ? T : size;
  for (uint32_t sz = size_lo; sz <= size_hi; sz <<= 1) {
    for (uint32_t stride = min(sz >> 1, T >> 1); stride > 0; stride >>= 1) {
      for (uint32_t k = threadIdx.x; k < (T >> 1); k += kSortThreads) {
        const uint32_t lo = 2u * k - (k & (stride - 1u)), hi = lo + stride;   // pair (lo, lo + stride) of the network
        const bool ascending = (((a0 + lo) & tile_mask) & sz) == 0u;          // (position inside the tile: the last stage, sz = the tile, is ascending everywhere)
        const uint64_t x = tile_keys[lo], y = tile_keys[hi];
        if ((x > y) == ascending) { tile_keys[lo] = y; tile_keys[hi] = x; }
      }
      __syncthreads();
    }
  }
  for (uint32_t k = threadIdx.x; k < T; k += kSortThreads) {
    const uint32_t i = a0 + k;
    const uint64_t v = tile_keys[k];
    if (!last) keys[i] = v;
    else if (i < n) { out_p[i] = (uint32_t)(v >> 32); out_dest[i] = (i & ~tile_mask) + (uint32_t)v; }
  }
}
// one pair per thread of stage `size`, stride ≥ a block: n_pad entries (a multiple of the tile)
__global__ __launch_bounds__(kBlock) void k_tile_merge_global(uint64_t* __restrict__ keys, uint64_t n_pairs, uint32_t stride, uint32_t size, uint32_t tile_mask) {
  for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < n_pairs; k += (uint64_t)gridDim.x * kBlock) {
    const uint64_t lo = 2ull * k - (k & (uint64_t)(stride - 1u)), hi = lo + stride;
    const bool ascending = (((uint32_t)lo & tile_mask) & size) == 0u;
    const uint64_t x = keys[lo], y = keys[hi];
    if ((x > y) == ascending) { keys[lo] = y; keys[hi] = x; }
  }
}

}  // namespace

void launch_fill_u32(uint32_t* p, uint64_t n, uint32_t v, hipStream_t s) { if (n) hipLaunchKernelGGL(k_fill_u32, grid_of(n), kBlock, 0, s, p, n, v); }
void launch_rank_scatter(const uint32_t* seq, uint32_t n_seq, const uint32_t* c2v, uint32_t* rank, hipStream_t s) {
  if (n_seq) hipLaunchKernelGGL(k_rank_scatter, grid_of(n_seq), kBlock, 0, s, seq, n_seq, c2v, rank);
}
// The face order of one mesh: order[j] = the face at position j, new_face[f] = its position.  Faces sorted by key (= the smallest sequence
// index among their vertices, none_key = n_keys - 1 for faces no coded vertex touches), equal keys in face order: a counting sort —
// bucket sizes by atomics, a prefix sum (dmi_conn.hip), placement, a tiny sort inside every bucket.  No library sort.
// count / fill: n_keys + 1 words each, zeroed by this call; scan_partials: scan_partials_words(n_keys + 1).
hipError_t launch_face_order(const uint32_t* c2v, const uint32_t* rank, uint32_t F, uint32_t n_keys, uint32_t* key, uint32_t* count, uint32_t* fill, uint32_t* scan_partials,
                             uint32_t* order, uint32_t* new_face, hipStream_t s) {
  if (!F) return hipSuccess;
  hipError_t e;
  if ((e = hipMemsetAsync(count, 0, ((size_t)n_keys + 1) * 4, s)) != hipSuccess) return e;
  (void)fill;   // (round 3's second round of atomics: the arrival numbers of the counting pass — parked in new_face until k_sort_buckets writes it — replace it)
  hipLaunchKernelGGL(k_face_keys, grid_of(F), kBlock, 0, s, c2v, rank, F, n_keys - 1, key, count, new_face);
  launch_exclusive_scan_u32(count, n_keys + 1, scan_partials, s);
  hipLaunchKernelGGL(k_place_faces, grid_of(F), kBlock, 0, s, key, F, count, new_face, order);
  hipLaunchKernelGGL(k_sort_buckets, grid_of(n_keys), kBlock, 0, s, count, n_keys, order, new_face);
  return hipSuccess;
}
void launch_remap_table(const uint32_t* c2v, const uint32_t* opp, const uint32_t* rank, const uint32_t* order, const uint32_t* new_face, uint64_t C, uint32_t* c2r_out, uint32_t* opp_out,
                        hipStream_t s) {
  if (C) hipLaunchKernelGGL(k_remap_table, grid_of(C), kBlock, 0, s, c2v, opp, rank, order, new_face, C, c2r_out, opp_out);
}
void launch_remap_seq(const uint32_t* seq, uint32_t n_seq, const uint32_t* new_face, const uint32_t* c2p, uint32_t* seq_out, uint32_t* s2p_out, hipStream_t s) {
  if (n_seq) hipLaunchKernelGGL(k_remap_seq, grid_of(n_seq), kBlock, 0, s, seq, n_seq, new_face, c2p, seq_out, s2p_out);
}
void launch_corner_ranks(const uint32_t* c2v, const uint32_t* rank, uint64_t C, uint32_t* c2r, hipStream_t s) { if (C) hipLaunchKernelGGL(k_corner_ranks, grid_of(C), kBlock, 0, s, c2v, rank, C, c2r); }
void launch_face_records(const uint32_t* c2v, const uint32_t* rank, const uint32_t* opp, uint32_t F, uint32_t* frec, hipStream_t s) {
  if (F) hipLaunchKernelGGL(k_face_records, grid_of(F), kBlock, 0, s, c2v, rank, opp, F, reinterpret_cast<uint4*>(frec));
}
void launch_rank_and_points(const uint32_t* seq, uint32_t n_seq, const uint32_t* c2v, const uint32_t* c2p, uint32_t* rank, uint32_t* s2p, hipStream_t s) {
  if (n_seq) hipLaunchKernelGGL(k_rank_and_points, grid_of(n_seq), kBlock, 0, s, seq, n_seq, c2v, c2p, rank, s2p);
}
void launch_compose_s2v(const uint32_t* s2p, uint32_t n_seq, const uint32_t* p2v, uint32_t num_points, uint32_t num_unique, uint32_t* s2v, uint32_t* bad, hipStream_t s) {
  if (n_seq) hipLaunchKernelGGL(k_compose_s2v, grid_of(n_seq), kBlock, 0, s, s2p, n_seq, p2v, num_points, num_unique, s2v, bad);
}
void launch_max_u32(const uint32_t* a, uint64_t n, uint32_t* out, hipStream_t s) { if (n) hipLaunchKernelGGL(k_max_u32, std::min<uint32_t>(grid_of(n), 2048u), kBlock, 0, s, a, n, out); }

size_t tile_sort_scratch_bytes(uint32_t n, uint32_t tile_log2, uint32_t local_log2) {
  if (tile_log2 <= local_log2) return 0;
  const uint64_t T = 1ull << tile_log2;
  return (size_t)(((uint64_t)n + T - 1) / T * T * 8);
}
hipError_t launch_tile_sort(const uint32_t* s2p, uint32_t n, uint32_t tile_log2, uint32_t local_log2, uint64_t* scratch, uint32_t* s2p_sorted, uint32_t* dest, hipStream_t s) {
  if (!n) return hipSuccess;
  if (local_log2 > kTileSortMaxLog2) local_log2 = kTileSortMaxLog2;
  if (tile_log2 < 6 || tile_log2 > 24 || local_log2 < 6) return hipErrorInvalidValue;
  // 128 KB of the CU's 160 (set on every call: the attribute belongs to the current device, and a process may drive several)
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_sort_local), hipFuncAttributeMaxDynamicSharedMemorySize, 8 << kTileSortMaxLog2);
  if (attr != hipSuccess) return attr;
  if (tile_log2 <= local_log2) {   // a tile is a block
    const uint32_t T = 1u << tile_log2;
    hipLaunchKernelGGL(k_tile_sort_local, (n + T - 1) / T, kSortThreads, 8u << tile_log2, s, s2p, n, tile_log2, tile_log2, 0u, 1, 1, (uint64_t*)nullptr, s2p_sorted, dest);
    return hipGetLastError();
  }
  if (!scratch) return hipErrorInvalidValue;
  const uint64_t T = 1ull << tile_log2, n_pad = ((uint64_t)n + T - 1) / T * T;
  const uint32_t B = 1u << local_log2, blocks = (uint32_t)(n_pad >> local_log2), lds = 8u << local_log2, tile_mask = (uint32_t)(T - 1);
  hipLaunchKernelGGL(k_tile_sort_local, blocks, kSortThreads, lds, s, s2p, n, local_log2, tile_log2, 0u, 1, 0, scratch, s2p_sorted, dest);
  for (uint32_t size = B << 1; size <= (uint32_t)T; size <<= 1) {
    for (uint32_t stride = size >> 1; stride >= B; stride >>= 1)
      hipLaunchKernelGGL(k_tile_merge_global, (uint32_t)std::min<uint64_t>((n_pad / 2 + kBlock - 1) / kBlock, 65535ull * 16), kBlock, 0, s, scratch, n_pad / 2, stride, size, tile_mask);
    hipLaunchKernelGGL(k_tile_sort_local, blocks, kSortThreads, lds, s, s2p, n, local_log2, tile_log2, size, 0, size == (uint32_t)T ? 1 : 0, scratch, s2p_sorted, dest);
  }
  return hipGetLastError();
}

}  // namespace dmi
