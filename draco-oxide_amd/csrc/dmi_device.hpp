// dmi_device.hpp — launch wrappers of the gfx950 kernels in dmi_kernels.hip.
// All pointers are device pointers unless stated otherwise; every launch goes to `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

#include <cstdint>

#include "dmi_debug.hpp"

namespace dmi {

// ---- kernel steps ---------------------------------------------------------------------------------
// Every data-parallel launch of the encode pipeline is described by a KernelStep (kernel id, argument block, grid).  The
// launch_* functions below either launch their step at once or — while a sink is set for the calling thread — append it to
// the sink, so that a batch driver can run the SAME phase of many jobs in one multi-item launch (launch_steps_multi).
// `level` orders the steps of one job: steps of equal level are independent of each other.
enum KernelId : int { K_RANGES, K_RANGES_FINAL, K_SEQ_QUANT, K_SEQ_QUANT_BIG, K_I32_FINAL, K_FUSED_PNU, K_FUSED_PN, K_FUSED_PU, K_FUSED_N,
                      K_PACKED_PNU, K_PACKED_PN, K_PACKED_PU, K_PACKED_N /* the same sweeps on packed values (QFmt) */,
                      K_WINDOW_PNU, K_WINDOW_PN, K_WINDOW_PU /* … with LDS-staged neighbourhoods */,
                      K_PAR1, K_PAR2, K_PAR3, K_PAR4, K_DELTA, K_TEX, K_ORIENT, K_HIST,
                      K_TEX_FIXUP, K_RANS_PREP, K_BITS_PREP, K_ORIENT_PREP, K_BATCH_FLAGS, K_TABLES /* tables + record prep: dmi_chains.hip */, K_COUNT };
constexpr int kStepLevels = 7;        // data-parallel phases: levels 0..6
constexpr int kPrepLevels = 3;        // tables (device form), record prep, batch flags: levels kStepLevels + 0..2 when planned together with the phases
struct KernelStep { int id; int level; uint32_t blocks; uint32_t lds; uint32_t args_size; uint32_t pad; alignas(8) uint8_t args[640]; };
void set_step_sink(std::vector<KernelStep>* sink);   // thread-local; nullptr = launch immediately
bool step_sink_push(const KernelStep& st);           // true = a sink is set and took the step
bool step_sink_active();
void launch_step(const KernelStep& st, hipStream_t s);
// items: device array of the kernel's argument blocks; block_info[b] = {item, block within the item}; item_blocks[i] = its grid
void launch_steps_multi(int id, const void* items, const uint2* block_info, const uint32_t* item_blocks, uint32_t total_blocks, uint32_t lds, hipStream_t s);

// ---- quantization (a4-a6), fused with the coding-order gather ---------------------------------------
// meta layout (floats): [0..N) per-component min, [N] range, [N+1..2N] per-component max (debug)
// Stage 1 for up to kMaxRangeAtts attributes in two launches: f32 min/max (kind 0, seeded 0.0: Q1) → meta, zero-length
// normal check (kind 1) → small[4]; every attribute's 16 scratch words are (re)initialised: [0..1] = {INT_MAX, INT_MIN},
// and `zero[0..zero_words)` (the rest of its slab slot: ranges, histogram, summaries) is cleared.
constexpr int kMaxRangeAtts = 8;
constexpr uint32_t kRangeMaxBlocks = 1024;   // partials: kRangeMaxBlocks * 2N floats per attribute
struct RangeAtt { const float* raw; float* partials; float* meta; uint32_t* small; uint32_t* zero; size_t zero_words; uint32_t n; int N; int kind; uint32_t first_block, blocks; };
struct RangeArgs { RangeAtt a[kMaxRangeAtts]; int count; };
void launch_value_ranges(RangeArgs& args, hipStream_t s);
// joint i32 min/max of the sequence-ordered quantized values: per-block partials of k_seq_quantize → minmax[2]
struct MinMaxAtt { const int32_t* ipartials; int32_t* minmax; uint32_t blocks; uint32_t pad; };
struct MinMaxArgs { MinMaxAtt a[kMaxRangeAtts]; int count; };
void launch_i32_minmax_final(const MinMaxArgs& args, hipStream_t s);
constexpr uint32_t kSeqQuantizeMaxBlocks = 8192;
constexpr uint32_t kSeqQuantizeBigEntries = 1u << 24;   // longer sequences: k_seq_quantize_big (dmi_kernels.hip)
uint32_t seq_quantize_blocks(uint32_t n);   // grid of launch_seq_quantize = partials written per attribute (≤ kSeqQuantizeMaxBlocks)
// qs[i] = portabilize(raw[s2v ? s2v[i] : s2p[i]]) for every attribute of one corner table (s2p[i] = point_idx(seq[i])) + per-block joint i32 min/max
// partials (ipartials: int32[2 * seq_quantize_blocks(n)]).  kind: 0 coordinate-wise, 1 octahedral, 2 ToBits.
constexpr int kMaxGather = 4;
// Layout of an attribute's quantized values `qs` (one entry per sequence index):
//   QF_I32  nq int32 per entry (any shape; every kernel reads it)
//   QF_P64  positions of a fused sweep, ≤ 21 bits: x | y << 21 | z << 42 in one uint64 — a neighbour is ONE aligned 8-byte gather (12-byte rows
//           straddle 128-byte lines) and the sweep's LDS windows hold 1.5× as many vertices
//   QF_B16  octahedral normals inside a fused sweep: u | v << 8 in a uint16        QF_H32  texture coordinates inside a fused sweep, ≤ 16 bits: u | v << 16
// Coordinate-wise and octahedral quantization only produce values in [0, 2^bits) (NaN → 0, ±inf → 0 or the range ends: see quant_coord), so the
// packed forms lose nothing.  Symbols are stored as uint16 when the attribute's alphabet bound (symbol_bins) fits 16 bits (`sym16`).
enum QFmt : int { QF_I32 = 0, QF_P64 = 1, QF_B16 = 2, QF_H32 = 3 };
struct QuantAtt { const float* raw; const uint32_t* s2v; void* qs; int32_t* ipartials; const float* meta; float maxq; int kind; int N; int fmt; };
struct QuantArgs { QuantAtt a[kMaxGather]; int count; };
// dest (nullable): the pass runs in TILE-SORTED order — slot j holds point s2p[j] and is written to sequence index dest[j]; inside a tile of
// consecutive sequence entries the slots are ordered by point, so a wavefront's gathers fall on a few lines instead of one line per lane
struct SeqQuantArgs { const uint32_t* s2p /* null: value order (entry i reads value i) */; uint32_t n; uint32_t pad; const uint32_t* dest; QuantArgs q; };
// The early stage of a whole-mesh call (dmi_kernels.hip): rec[v] = the quantized position / normal / texture coordinate of value v in one 16-byte record
// (value order; nrm / uv null: the mesh has none), then qs_*[i] = the fields of rec[s2p[i]] + per-block joint i32 min/max partials per attribute
// (indices 0 / 1 / 2 of the per-attribute arrays below = position / normal / texture coordinate; null = the mesh has none)
// range partials: the per-block pairs of launch_value_range_partials (k_value_ranges without its `_final`): every block of k_value_quantize_rec folds them
// itself, block 0 also writes the attributes' slots ([small 16 words][meta 16 words], as k_value_ranges_final leaves them)
struct ValueRecArgs { const float *pos, *nrm, *uv; const float *pos_partials, *uv_partials; uint32_t* nrm_flags /* out: a zero-length normal seen, one word per block */; uint32_t range_blocks[3]; float pos_maxq, uv_maxq; uint32_t n; void* rec;
                      uint32_t* slot[3]; int32_t* ipartials[3] /* 2 * value_quantize_rec_blocks(n) words each */; };
// the stage's slots → the job's slab slots, the joint i32 min/max folded from the per-block pairs into words 0–1 (the first block of the consumer kernel)
struct EarlySlots { const uint32_t* src[3]; uint32_t* dst[3]; const int32_t* ipartials[3]; const uint32_t* nrm_flags /* per block of the quantizer: folded into word 4 of the normal's slot */; uint32_t ipartial_blocks; uint32_t pad; };
struct GatherRecArgs { const uint32_t* s2p; uint32_t n; uint32_t pad; const void* rec; uint64_t* qs_pos; uint16_t* qs_nrm; uint32_t* qs_uv; EarlySlots slots; };
// (measured, 10M triangles: 512 blocks per attribute 38.4 µs + 46.2 for the quantizer that folds them, 768: 34.1 + 47.0, 1024: 32.2 + 48.4 with two values per thread and round)
constexpr uint32_t kEarlyRangeBlocks = 1024;   // blocks per attribute of the early stage's range pass = partial pairs every block of k_value_quantize_rec folds
void launch_value_range_partials(RangeArgs& args, uint32_t max_blocks, hipStream_t s);   // k_value_ranges alone: sets a.blocks / a.first_block
uint32_t value_quantize_rec_blocks(uint32_t n);
void launch_value_quantize_rec(const ValueRecArgs& a, hipStream_t s);
void launch_seq_gather_rec(const GatherRecArgs& g, hipStream_t s);
void launch_seq_quantize(const uint32_t* s2p, const uint32_t* dest, uint32_t n, const QuantArgs& args, hipStream_t s);

// Fan-row sweep (see k_predict_fused).  Seam-free fast path: position (parallelogram, 3 components) + normal and/or texture
// coordinates coded on the SAME corner table in one sweep; qs_nrm / qs_uv null = attribute absent.  sym_pos null = a normal
// attribute alone on its own table (c2r = the position table's array, fan rows built with centre_in_apex).
struct FusedArgs {
  const uint32_t* seq; const uint32_t* c2r; const uint32_t* opp; uint32_t n;
  uint32_t packed;   // 1: qs_pos is QF_P64, qs_nrm QF_B16, qs_uv QF_H32; 0: all QF_I32
  const void* qs_pos; const int32_t* mm_pos; void* sym_pos;
  const void* qs_nrm; void* sym_nrm; uint8_t* flips; uint32_t* counters;
  uint32_t* flip_partials;   // one word per block of the launch: its count of unflipped normals (summed into counters[0] by the histogram launch — a same-address atomic per block costs ≈ 11 ns each, serialised: 90 µs for 8192 blocks)
  const void* qs_uv; const int32_t* mm_uv; void* sym_uv; uint8_t* orient;
  const uint32_t* fan_hdr; const uint32_t* fan_apex; const uint32_t* fan;   // fan rows of the table (launch_build_fans)
  uint32_t sym16;    // bit 0 / 1 / 2: sym_pos / sym_nrm / sym_uv are uint16 arrays
  uint32_t face_stride;   // 0 / 3: c2r and opp are dense arrays indexed by corner; 8: face records (launch_face_records) — c2r = the records, opp = c2r + 4, entry of
                          // corner c at 8·(c / 3) + c % 3.  Only rows that overflow (valence > 8) and deferred texture coordinates read them.
  // Texture-coordinate entries outside the sweep's exact f64 tier (large operands) are not predicted in the sweep: their sequence indices
  // go to fix_list (fix_count[0] of them, appended one atomic per wavefront) and k_texcoord_fixup — launched right after the sweep —
  // predicts them with the general i64 form.  The sweep itself then holds no out-of-line call: 58 VGPRs instead of 80, 8 waves per SIMD.
  uint32_t* fix_list; uint32_t* fix_count;
};
void launch_predict_fused(const FusedArgs& a, hipStream_t s);
struct ParArgs { const uint32_t* seq; const uint32_t* c2r; const uint32_t* opp; const int32_t* qs; const int32_t* minmax; void* sym; uint32_t n; uint32_t sym16; };
struct DeltaArgs { uint64_t n_comp; const int32_t* qs; void* sym; int N; int sym16; };
struct TexArgs { const uint32_t* seq; const uint32_t* c2r; const int32_t* qs; const uint32_t* c2r_pos; const void* qs_pos /* QF_I32 or QF_P64: pos_fmt */; const int32_t* minmax; void* sym;
                 uint8_t* orient; uint32_t n; uint32_t sym16; int pos_fmt; int pad; };
struct OrientArgs { const uint8_t* orient; uint32_t* summary; uint32_t n; uint32_t pad; };
// Per coded vertex, in coding order: hdr[n], apex[n], fan[8n] (32-byte aligned) — see k_build_fans.  Once per job.
// centre_in_apex: apex[i] = c2r[seq[i]] (the fan centre's rank) instead of the rank across the opposite edge — for a normal
// attribute swept on its own table (c2r = the position table's, opp/seq = the normal table's).
void launch_build_fans(const uint32_t* seq, uint32_t n, const uint32_t* c2r, const uint32_t* opp, uint32_t* hdr, uint32_t* apex, uint32_t* fan, bool centre_in_apex,
                       hipStream_t s);

// ---- predict + transform (a7-a14) → symbols ------------------------------------------------------
// c2r[c] = sequence index of corner c's vertex (DMI_NONE if never coded): "already coded" ⇔ c2r[c] < i
void launch_pred_parallelogram_wrapped(const uint32_t* seq, uint32_t n, const uint32_t* c2r, const uint32_t* opp,
                                       const int32_t* qs, const int32_t* minmax, int N, void* sym, bool sym16, hipStream_t s);
void launch_pred_delta_difference(uint32_t n, const int32_t* qs, int N, void* sym, bool sym16, hipStream_t s);
// orient[i]: 0 = no bit pushed, 1 = false, 2 = true
void launch_pred_texcoord_wrapped(const uint32_t* seq, uint32_t n, const uint32_t* c2r, const int32_t* qs, const uint32_t* c2r_pos,
                                  const void* qs_pos, int pos_fmt, const int32_t* minmax, void* sym, bool sym16, uint8_t* orient, hipStream_t s);
// per-block summaries of the orientation flags, stitched on the host:
// summary[b] = {valid_count, first_value(0/1, 2 = none), last_value, internal_transitions}
void launch_orient_summary(const uint8_t* orient, uint32_t n, uint32_t* summary, uint32_t* n_blocks_out_host, hipStream_t s);
uint32_t orient_summary_blocks(uint32_t n);

// ---- histogram (a16) -------------------------------------------------------------------------------
// histograms (pre-zeroed) of up to kMaxRangeAtts symbol streams in one launch; *overflow |= 1 when a symbol ≥ bins is met
struct HistAtt { const void* sym; uint64_t n; uint32_t* hist; uint32_t* overflow; uint32_t bins; uint32_t first_block, blocks, sym16;
                 const uint32_t* flip_partials; uint32_t* flip_count; uint32_t n_flip_partials, pad; };   // a normal attribute: its sweep's per-block counts → flip_count[0]
// `orient` (optional, orient.orient != null): the orientation-flag summaries of one texture-coordinate attribute, computed by extra
// blocks of the same launch (launch_orient_summary is the standalone form)
struct HistArgs { HistAtt a[kMaxRangeAtts]; int count; uint32_t hist_blocks; OrientArgs orient; };
void launch_histograms(HistArgs& args, hipStream_t s);
constexpr uint32_t kSweepMaxBlocks = 16384;   // grid cap of the fused predictor sweep = entries of a flip_partials array
uint32_t predict_fused_blocks(uint32_t n);    // grid of launch_predict_fused

// ---- serial coders: two wavefronts per stream (a18, a19, a11, a13) ------------------------------------
// Coding record of one symbol (see dmi_chains.hip), 20 bytes: x / f = mulhi(x, m) >> (b & 31); bit 8 of b flags f == 1, bit 9
// f < 2^(P-8); d = 2^P - f; c = cumulative frequency; t = renormalisation threshold (x ≥ t ⇒ at least one byte leaves).
struct RansEntry { uint32_t m, b, d, c, t; };   // t = f << 10 (rANS) / f << 12 (rABS): the renormalisation threshold
constexpr size_t kChainPad = 384;   // records a stream's buffer extends past n: the chain's look-ahead loads may run that far
// Coding record of a symbol with normalised frequency f (see dmi_chains.hip for the exactness argument); precision 8 = rABS.
__host__ __device__ inline RansEntry make_rans_entry(uint32_t f, uint32_t cum, uint32_t precision) {
  RansEntry e{0u, 0u, 0u, cum, 0u};
  if (f == 0) return e;                       // never coded
  e.t = f << (precision == 8 ? 12 : 10);      // rans.rs:40 `state >= (L >> P) * f << 8` with L = 4·2^P; rABS :97 with L = 4096
  e.d = (1u << precision) - f;
  // bit 9 of b: f < 2^(P-8) — the state can exceed f·2^18, i.e. this symbol may renormalise by more than one byte
  const uint32_t multi = (precision >= 8 && ((uint64_t)f << 8) < ((uint64_t)1 << precision)) ? 0x200u : 0u;
  if (f == 1) { e.m = 0xFFFFFFFFu; e.b = 0x100u | multi; return e; }   // flagged: the batch takes the generic loop
  const unsigned lg = 31u - (unsigned)__builtin_clz(f);
  if ((f & (f - 1)) == 0) { e.m = 0x80000000u; e.b = (lg - 1) | multi; return e; }
  e.m = (uint32_t)((((uint64_t)1 << (32 + lg)) + f - 1) / f);
  e.b = lg | multi;
  return e;
}
// batch_flags: (n + 63) / 64 + 1 words; [b] != 0 ⇔ batch b holds a frequency-1 symbol
void launch_rans_prep(const void* sym, bool sym16, uint64_t n, const RansEntry* table, uint32_t bins, RansEntry* rec, uint32_t* batch_flags, hipStream_t s);
void launch_batch_flags(const RansEntry* rec, uint64_t n, const uint32_t* n_dev, uint32_t* batch_flags, hipStream_t s);
void launch_bits_prep(const uint8_t* bits, uint64_t n, RansEntry e0, RansEntry e1, RansEntry* rec, hipStream_t s);
void launch_bits_prep_dev(const uint8_t* bits, uint64_t n, const RansEntry* entries, RansEntry* rec, hipStream_t s);   // record pair in device memory
// chunk_info[2c] = compact offset of 4096-flag chunk c, [2c+1] = value (0/1) of the first valid flag after it (1 if none)
void launch_orient_prep(const uint8_t* orient, uint32_t n, const uint32_t* chunk_info, RansEntry e0, RansEntry e1, RansEntry* rec, hipStream_t s);
void launch_orient_prep_dev(const uint8_t* orient, uint32_t n, const uint32_t* chunk_info, const RansEntry* entries, RansEntry* rec, hipStream_t s);
// the compacted transition bits (bits_out[k] = bit k of the stream, 1 byte each) instead of coding records: the stream is coded on a host core
void launch_orient_bits(const uint8_t* orient, uint32_t n, const uint32_t* chunk_info, uint8_t* bits_out, hipStream_t s);
struct ChainDesc {
  uint32_t kind;            // 0 = rANS over coding records, 1 = rABS over flips (forward), 2 = rABS over orientation flags
  uint32_t precision;       // rANS precision bits
  uint64_t n;               // symbols / entries
  const void* sym;          // unused by the kernel (kept for debugging)
  uint32_t one_byte;        // kind 0: 1 = almost every batch is free of rare symbols (f < 2^(P-8)): take the one-byte-renormalisation step
  uint32_t pad0;
  const RansEntry* table;   // kind 0: n coding records in coding order (k_rans_prep output)
  const uint32_t* batch_flags;   // kind 0: per-batch frequency-1 flags (k_rans_prep), NULL otherwise
  uint32_t force_generic;   // kind 1/2: 1 when a coded frequency is 1 (p0 ∈ {1, 255}): every batch takes the generic loop
  uint32_t state0;          // initial state: 4 << precision (rANS) / 4096 (rABS)
  uint8_t* out;             // byte output
  uint64_t cap;
  uint32_t* out_len;        // [0] = bytes written, [1] = error flag (1 = state too large, 2 = capacity)
  uint32_t* ticks;          // optional: chain duration in 100 MHz ticks
};
// Persistent chain kernel: `order_dev` (nullable) = stream indices longest first; `next_stream_dev` = one device word for the
// pull counter (zeroed by the launcher on the stream).
uint32_t chain_grid(uint32_t n_streams);
// sparse = sparse parking on two pairs per CU (see k_chains); chain_launch_sparse picks the form from the streams' step counts
bool chain_launch_sparse(uint64_t longest_steps, uint64_t total_steps, uint32_t n_streams);
void launch_chains(const ChainDesc* descs_dev, const uint32_t* order_dev, uint32_t n_streams, uint32_t* next_stream_dev, bool sparse, hipStream_t s);
// argument blocks of the record-prep kernels (K_RANS_PREP … K_BATCH_FLAGS); launch_step / launch_steps_multi dispatch to these
// `bins`: symbols ≥ bins (only possible after a histogram overflow, which is reported as an error) take an all-zero record.
// `entries` (nullable): {record of bit 0, record of bit 1} in device memory (written by k_tables) instead of e0 / e1;
// `n_dev` (nullable): the record count in device memory (texture-coordinate orientation streams, whose length the device finds).
struct RansPrepArgs { const void* sym; const RansEntry* table; RansEntry* rec; uint32_t* batch_flags; uint64_t n; uint32_t bins; uint32_t sym16; };
struct BitsPrepArgs { const uint8_t* bits; RansEntry* rec; uint64_t n; RansEntry e0, e1; const RansEntry* entries; };
struct OrientPrepArgs { const uint8_t* orient; const uint32_t* chunk_info; RansEntry* rec; RansEntry e0, e1; uint32_t n; uint32_t pad; const RansEntry* entries;
                        uint8_t* bits_out; /* nullable: the compacted transition bits themselves (1 byte each) instead of coding records — host-core chains */ };
struct BatchFlagsArgs { const RansEntry* rec; uint32_t* batch_flags; uint64_t n; const uint32_t* n_dev; };
// Device form of the table stage (a17 + the descriptors of the streams): one workgroup per attribute normalises the histogram
// (RansSymbolEncoder::new, rans.rs:146-239), serialises the table, builds the coding-record table, closes the metadata
// streams' parameters (zero_prob, record pair, orientation chunk offsets) and writes the attribute's chain descriptors —
// nothing of an encode comes back to the host before the chains have run.
// small[] words written here: [6] header bytes, [7] table error (1 empty histogram, 2 normalisation overflow, 3 underflow,
// 4 occurring symbol normalised to 0, 5 header capacity), [12] rANS precision (until a device chain stores its clock there),
// [14] metadata zero_prob, [15] metadata entry count.
struct TableAtt {
  const uint32_t* hist; uint32_t* freq /* scratch: bins words */; RansEntry* rtable; uint8_t* hdr; uint32_t* small;
  uint64_t n_sym; uint32_t bins; uint32_t hdr_cap;
  ChainDesc* desc; const void* sym; const RansEntry* rec; const uint32_t* batch_flags; uint8_t* out; uint64_t out_cap;   // the rANS stream
  uint32_t aux_kind /* 0 none, 1 normal flips, 2 texture-coordinate orientations */; uint32_t n_entries /* sequence entries */;
  const uint32_t* summary; uint32_t* chunk_info; RansEntry* aux_entries; uint32_t summary_blocks; uint32_t pad;
  ChainDesc* aux_desc; const RansEntry* aux_rec; const uint32_t* aux_flags; uint8_t* aux_out; uint64_t aux_cap;
  ChainDesc* hdr_desc;   // nullable: pseudo-descriptor {out = hdr, out_len = small + 6} so that the header rides the packed read-back
};
constexpr uint32_t kTablesThreads = 1024;
void launch_tables(const TableAtt& a, hipStream_t s);
constexpr int kTableGroup = 8;
struct TableGroup { TableAtt a[kTableGroup]; int count; int pad; };
void launch_tables_group(const TableGroup& g, hipStream_t s);   // immediate launch (no step sink): one job's attributes, one block each
void launch_prep_step(const KernelStep& st, hipStream_t s);
void launch_prep_steps_multi(int id, const void* items, const uint2* block_info, const uint32_t* item_blocks, uint32_t total_blocks, hipStream_t s);
// Batch read-back: pack the coded bytes of every stream into `arena` (16-byte aligned slots, in stream order);
// table[k] = {offset, length, error} and table[n_streams].offset = total bytes.  Stream outputs must be allocated with
// ≥ 16 bytes of slack (the copy moves whole 16-byte words).
struct PackEntry { uint64_t offset; uint32_t len; uint32_t err; };
// items[k] = {device source (16-byte aligned), destination offset in `arena`, bytes (multiple of 16)}
struct CopyItem { const void* src; uint64_t dst_offset; uint64_t bytes; };
void launch_copy_items(const CopyItem* items_dev, uint32_t n_items, uint8_t* arena, hipStream_t s);
// the inverse: arena[dst_offset .. +bytes) → items[k].src (a device destination)
void launch_scatter_items(const CopyItem* items_dev, uint32_t n_items, const uint8_t* arena, hipStream_t s);
// items[k].src .. +bytes (a device range, a multiple of 4 bytes) := 0, every range of a batch in one launch
void launch_clear_items(const CopyItem* items_dev, uint32_t n_items, hipStream_t s);
// up to kClearRanges ranges (4-byte aligned starts, any length) filled with a byte value each (0 by default) in ONE launch, the list passed by value (no
// descriptor upload): the hipMemsetAsync calls in front of a stage's kernels were a launch of ≈ 5–10 µs each — 235 of them per 1024-file transcode
constexpr uint32_t kClearRanges = 24;
struct ClearRanges {
  void* p[kClearRanges]; uint64_t bytes[kClearRanges]; uint8_t value[kClearRanges]; uint32_t count; uint32_t pad;
  void add(void* ptr, uint64_t n, uint8_t v = 0) { if (n && count < kClearRanges) { p[count] = ptr; bytes[count] = n; value[count] = v; ++count; } }
  bool full() const { return count == kClearRanges; }
};
void launch_clear_ranges(const ClearRanges& r, hipStream_t s);
void launch_pack_streams(const ChainDesc* descs_dev, uint32_t n_streams, PackEntry* table, uint8_t* arena, hipStream_t s);

// ---- decoder side: the data-parallel stages of reading an attribute section back (dmi_decode.cpp drives them) ----
// normals: entry i (corner seq[i] of the normal attribute's table) is predicted from the decoded positions of its fan and corrected by its
// two symbols; the octahedral coordinates land at the attribute vertex of the corner
struct DecodeNormalArgs { const uint32_t* seq; uint32_t n; uint32_t pad; const uint32_t* c2v_pos /* universal vertex per corner */; const uint32_t* opp /* this table's */;
                          const uint32_t* c2v_att; const int32_t* pos_by_vertex /* 3 per universal vertex */; const uint32_t* sym; const uint8_t* flips; int32_t* oct_by_vertex; };
void launch_decode_normals(const DecodeNormalArgs& a, hipStream_t s);
// values[point_idx(c)] = dequantize(q[vertex(c)]) for every corner; kind: 1 ToBits, 2 coordinate-wise (mn, delta), 3 octahedral
// last_corner[p] = 1 + the largest corner that references point p (launch_last_corners): a point met by several vertices of the attribute's
// table (a non-manifold vertex split in two) takes the value of its last corner — the result of a serial loop over the corners
struct DequantizeArgs { const uint32_t* c2p; const uint32_t* c2v; uint64_t corners; const int32_t* q; float* out; const uint32_t* last_corner; float mn[4]; float delta; int kind; int N; int pad; };
void launch_dequantize(const DequantizeArgs& a, hipStream_t s);
void launch_last_corners(const uint32_t* c2p, uint64_t corners, uint32_t* last_corner /* zeroed, one word per point */, hipStream_t s);

// ---- coding-order relabelling of the connectivity inputs on the device (dmi_relabel.hip; job creation of large meshes) ----
void launch_fill_u32(uint32_t* p, uint64_t n, uint32_t v, hipStream_t s);
void launch_rank_scatter(const uint32_t* seq, uint32_t n_seq, const uint32_t* c2v, uint32_t* rank /* pre-filled with DMI_NONE */, hipStream_t s);
// Tile-sorted quantize gather (job creation): slots of every 2^tile_log2-entry tile of the sequence ordered by point index; s2p_sorted[j] = the point
// slot j reads, dest[j] = the sequence entry it writes.  Tiles up to 2^local_log2 ≤ 2^kTileSortMaxLog2 entries are sorted by one workgroup in LDS; larger
// ones need `scratch` (tile_sort_scratch_bytes: one 8-byte key per entry of the padded sequence) for the long strides of the network.
constexpr uint32_t kTileSortMaxLog2 = 14;
size_t tile_sort_scratch_bytes(uint32_t n, uint32_t tile_log2, uint32_t local_log2);
hipError_t launch_tile_sort(const uint32_t* s2p, uint32_t n, uint32_t tile_log2, uint32_t local_log2, uint64_t* scratch, uint32_t* s2p_sorted, uint32_t* dest, hipStream_t s);
hipError_t launch_face_order(const uint32_t* c2v, const uint32_t* rank, uint32_t F, uint32_t n_keys, uint32_t* key, uint32_t* count, uint32_t* fill, uint32_t* scan_partials,
                             uint32_t* order, uint32_t* new_face, hipStream_t s);
void launch_remap_table(const uint32_t* c2v, const uint32_t* opp, const uint32_t* rank, const uint32_t* order, const uint32_t* new_face, uint64_t C, uint32_t* c2r_out, uint32_t* opp_out,
                        hipStream_t s);
void launch_remap_seq(const uint32_t* seq, uint32_t n_seq, const uint32_t* new_face, const uint32_t* c2p, uint32_t* seq_out, uint32_t* s2p_out, hipStream_t s);
void launch_compose_s2v(const uint32_t* s2p, uint32_t n_seq, const uint32_t* p2v, uint32_t num_points, uint32_t num_unique, uint32_t* s2v, uint32_t* bad, hipStream_t s);
void launch_max_u32(const uint32_t* a, uint64_t n, uint32_t* out /* pre-zeroed */, hipStream_t s);


// ---- connectivity tables on the device (dmi_conn.hip): the order-free half of the connectivity stage, batched ----
// M meshes concatenated: corner arrays are indexed by global corner (3·face_off[m] + local corner), vertex arrays by global vertex
// (vert_off[m] + local vertex, Vcap[m] slots per mesh = the Position attribute's value count); the ids STORED are mesh-local.
struct ConnMeshDesc { uint32_t face_off, vert_off, F, Vcap; uint32_t p2v_off /* into ConnArgs::p2v, DMI_NONE = identity */, num_points, pad0, pad1; };
enum ConnFlag : uint32_t {
  CONN_BAD_INDEX = 1,          // a face index ≥ num_points or a position value index ≥ Vcap → DMI_ERR_INVALID_ARGUMENT
  CONN_DEGENERATE = 2,         // a vertex-degenerate face        ┐
  CONN_NONMANIFOLD_EDGE = 4,   // an edge with more than two faces ├ the result depends on the corner order: the host runs the reference's serial walks
  CONN_MULTI_FAN = 8,          // a vertex with several fans       ┘
  CONN_UNUSED_VERTEX = 16,     // an id below the largest referenced one that no face uses (mod.rs:105-108 panic) → DMI_ERR_UNUSED_VERTICES
  CONN_HAS_BOUNDARY = 32       // some corner has no opposite (information: the Edgebreaker's boundary labelling can be skipped without it)
};
struct ConnArgs {
  const ConnMeshDesc* meshes; uint32_t M, total_faces, total_verts, pad;
  const uint32_t* faces;   // 3·total_faces point ids
  const uint32_t* p2v;     // concatenated position maps (nullable)
  uint32_t* c2v;           // out: vertex per corner; == faces when no mesh has a position map (then not written)
  uint32_t* opp;           // out
  uint32_t* lmc;           // out: per vertex
  uint8_t* on_boundary;    // out: per vertex
  uint32_t* flags;         // out: M words of ConnFlag
  uint32_t* vmax;          // out: M words, the largest vertex id a face references (V = vmax + 1)
  uint32_t *ecount, *efill, *first;   // scratch: total_verts + 1 words each
  uint32_t *he_key, *he_corner;       // scratch: 3·total_faces words each
  uint8_t* cdone;                     // scratch: 3·total_faces bytes
  uint32_t* scan_partials;            // scratch: scan_partials_words(total_verts + 1)
};
hipError_t conn_tables_clear(const ConnArgs& a, hipStream_t s);   // the memsets launch_conn_tables expects
void launch_conn_tables(const ConnArgs& a, hipStream_t s);
void launch_opp_quad(const uint32_t* opp, uint64_t C, uint32_t* out, hipStream_t s);   // out[c] = opp[c] as a 4·face + k id (kNone stays)
// Attribute corner tables of a batch on the device (core/corner_table/attribute_corner_table.rs:16-137), after launch_conn_tables on the same
// stream.  An "att item" = one (mesh, non-position attribute) whose point → value map may differ from the position's: seam test per edge
// (:44-63), attribute vertices per universal vertex = 1 + the seam edges its right swing crosses (:116-133), ids by a prefix sum, seam-aware fan
// starts (:101-113).  Per-corner arrays of the items are concatenated (corner_off), per-vertex scratch too (vert_off, Vcap slots per item);
// att vertex ids are item-local.  Items of meshes the universal kernels flagged are skipped (info stays 0: the host builds those meshes).
struct AttItemDesc { uint32_t mesh /* index of its ConnMeshDesc */, map_off /* words into ConnArgs::p2v, DMI_NONE = identity */, corner_off, vert_off; };
struct AttInfo { uint32_t interior /* some edge with two faces is a seam */, num_vertices, done /* 1: tables written */, pad /* offset of the item's left-most corners in AttArgs::lmc */; };
struct AttArgs {
  const AttItemDesc* items; uint32_t n_items, total_corners, total_verts, pad;
  uint8_t* seam;        // total_corners: 1 = the edge opposite the corner is a seam of the attribute (boundary edges included); zeroed by att_tables_clear
  uint8_t* vseam;       // total_verts: the universal vertex lies on a seam; zeroed
  uint32_t* count;      // total_verts + 1: attribute vertices per universal vertex → exclusive scan (zeroed)
  uint32_t* c2v;        // total_corners: attribute vertex per corner (item-local ids)
  uint32_t* opp;        // total_corners: opposite corner, DMI_NONE across seams
  uint32_t* lmc;        // total_corners (worst case): left-most corner per attribute vertex, at [Σ attribute vertices of the items before + id]
  AttInfo* info;        // n_items, zeroed
  uint32_t* scan_partials;   // scan_partials_words(total_verts + 1)
};
hipError_t att_tables_clear(const AttArgs& t, hipStream_t s);
void launch_att_tables(const ConnArgs& a, const AttArgs& t, hipStream_t s);
// Coding-order relabelling of a batch of meshes in one launch per step (the arrays of dmi_relabel.hip, per mesh): inputs in the mesh's own
// numbering (device arrays of the connectivity stage + the uploaded sequence), outputs in the job's memory.
// An item is one corner table of a job.  The face order is the universal table's (sequence.rs walks that one first): an attribute table of
// its own (interior seams: attribute_corner_table.rs:16-137) is relabelled with the universal item's face order (`order_item`) and its own
// vertex ranks; it takes no part in the counting sort (its slices of key / count are empty).
struct RelabelItem {
  const uint32_t *c2p, *c2v, *opp, *seq;
  uint32_t F, V, n_seq, order_item /* index of the item whose face order this table follows (itself for a universal table) */;
  uint32_t face_off, vert_off, key_off /* Σ (n_seq + 1) */, seq_off;   // this table's slices of the batch scratch arrays (face_off / key_off: universal tables only)
  uint32_t remap_off /* Σ F of the items before: the corner space of k_rl_remap */, plain /* 1: the job keeps the mesh's own face order (a one-shot job whose attributes all ride the fused sweep: DESIGN §3) — ranks, c2r and s2p only */, pad1, pad2;
  uint32_t *c2r, *opp_out, *seq_out, *s2p;
};
struct RelabelBatch {
  uint32_t any_sorted = 1;   // 0: every item is plain — the counting-sort launches are skipped
  const RelabelItem* items; uint32_t n_items, total_faces /* universal tables */, total_verts, total_keys, total_seq, total_remap_faces /* all tables */;
  uint32_t* rank;          // total_verts, filled with DMI_NONE
  uint32_t* key;           // total_faces
  uint32_t* count;         // total_keys + 1, zeroed: bucket sizes → starts (the arrival number the counting atomic returns is parked in new_face)
  uint32_t *order, *new_face;   // total_faces
  uint32_t* scan_partials;      // scan_partials_words(total_keys + 1)
};
void launch_relabel_batch(const RelabelBatch& b, hipStream_t s);
struct ComposeItem { const uint32_t* s2p; const uint32_t* p2v; uint32_t* s2v; uint32_t off /* Σ n_seq of the items before */, n; };
void launch_corner_ranks(const uint32_t* c2v, const uint32_t* rank, uint64_t C, uint32_t* c2r, hipStream_t s);   // c2r[c] = rank[c2v[c]]
// frec[8f..8f+2] = rank[c2v[3f..3f+2]], frec[8f+4..8f+6] = opp[3f..3f+2] (8F words): ranks and opposite corners of a face in one 32-byte record
void launch_face_records(const uint32_t* c2v, const uint32_t* rank, const uint32_t* opp, uint32_t F, uint32_t* frec, hipStream_t s);
// fan rows from face records (corner ids in seq / the records' opposite corners stay 3·face + k)
void launch_build_fans_rec(const uint32_t* seq, uint32_t n, const uint32_t* frec, uint32_t* hdr, uint32_t* apex, uint32_t* fan, hipStream_t s);
void launch_rank_and_points(const uint32_t* seq, uint32_t n_seq, const uint32_t* c2v, const uint32_t* c2p, uint32_t* rank, uint32_t* s2p, hipStream_t s);   // rank[c2v[seq[k]]] = k and s2p[k] = c2p[seq[k]] in one pass over the sequence
void launch_compose_batch(const ComposeItem* items_dev, uint32_t n_items, uint32_t total, hipStream_t s);
// fan rows of many tables in one launch (k_build_fans per item)
struct FanItem { const uint32_t *seq, *c2r, *opp; uint32_t *hdr, *apex, *fan; uint32_t off /* Σ n of the items before */, n, centre_in_apex, pad; };
void launch_build_fans_batch(const FanItem* items_dev, uint32_t n_items, uint32_t total, hipStream_t s);
void launch_exclusive_scan_u32(uint32_t* data, uint32_t n, uint32_t* partials, hipStream_t s);   // in place
size_t scan_partials_words(uint32_t n);

// ---- MeshBuilder::build on the device (dmi_build.hip): value dedup, point merge, degenerate faces, unused points — batched ----
// M meshes concatenated.  "Item" = one (mesh, attribute); att-point arrays are indexed by ap = item.ap_off + point, point arrays by
// mesh.point_off + point, face arrays by mesh.face_off + face.  Rows are 1–4 four-byte words.
constexpr uint32_t kMbMaxAtts = 16;   // (round 5: 8 → 16; a glTF primitive brings POSITION / NORMAL / TEXCOORD_0 and its _FEATURE_ID_n sets)
enum MbFlag : uint32_t { MB_BAD_INDEX = 1 /* a face index ≥ the point count */, MB_EMPTY = 2 /* no face survives: builder.rs:129 skips the point removal */,
                         MB_CROWDED = 4 /* a hash probe sequence ran past kMaxProbes (rows crafted to collide): the host builder takes the mesh */ };
struct MbMesh { uint32_t index, n_items, item0, P, F, face_off, point_off, ptab_off, ptab_mask, pad0, pad1, pad2; };
struct MbItem { uint32_t mesh, P, words, is_float, row_off /* words into raw_values */, ap_off, tab_off, tab_mask, stride /* words between rows (= words when packed) */, pad; };
struct MbMeshOut { uint32_t flags, nv /* largest referenced point + 1 */, F_out, P_out, face_out_off /* faces before this mesh's in arena A */, classes, pad0, pad1; };
struct MbItemOut { uint32_t n_first /* values after Attribute::from */, n_out /* values left at the end */, has_map, map_off /* words into arena A */, val_off /* words into arena B */, pad0, pad1, pad2; };
struct MbWiden { uint32_t off /* Σ elements of the items before */, bytes /* 1 or 2 */, dst_off /* words */, pad; uint64_t src_byte_off; };
struct MbArgs {
  const MbMesh* meshes; const MbItem* items; uint32_t M, n_items, total_faces, total_points, total_ap, pad;
  const uint32_t* raw_values;   // rows of every item
  const uint32_t* raw_faces;    // 3·total_faces point ids
  uint32_t *vtab, *ptab;        // hash tables (filled with DMI_NONE)
  uint32_t *vslot /* table slot of the row, then its byte class (k_mb_value_first) */, *vflag /* total_ap + 1 */, *vid, *vfirst, *vused /* total_ap + 1 */;
  uint32_t *pslot, *prep, *pflag /* total_points + 1 */, *used /* total_points + 1 */;
  uint32_t *keep /* total_faces + 1 */, *tmp_faces /* 3·total_faces */, *scan_partials;
  MbMeshOut* mesh_out; MbItemOut* item_out; uint32_t* totals /* [0] arena A words, [1] arena B words, [2] face words */;
  uint32_t *arena_a, *arena_b;
};
hipError_t mesh_build_clear(const MbArgs& a, size_t vtab_words, size_t ptab_words, hipStream_t s);
void launch_mesh_build(const MbArgs& a, hipStream_t s);
void launch_widen_indices(const MbWiden* items_dev, uint32_t n_items, uint32_t total, const uint8_t* src, uint32_t* dst, hipStream_t s);

}  // namespace dmi
