// oracle/oracle.hpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (C++17, single thread, no dependencies) of the reearth/draco-oxide encoder,
// written from the reference sources cited next to every function (paths are relative to
// /root/reference/draco-oxide/src/).  It exists only so that tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg can check / time the HIP product path against it.
// Nothing under draco-oxide_amd/ may include, link or call anything in this directory.
//
// PARITY STATUS: "parity unpinned" at the .drc byte level — the reference (Rust) cannot be
// built in this image (no cargo/rustc) and its own tests hold no golden .drc bytes
// (tests/compatibility.rs:7-16 asserts nothing).  The restatement is pinned against every
// known-answer test the reference does hold for this path (SURVEY.md §4 / tests/test_oracle_kat.py):
// traversal order, seam tables, corner tables, OBJ indexing, dedup maps, BitWriter bytes,
// LEB128, rANS/rABS round trips.
#pragma once
#include <array>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace orc {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;
using i64 = int64_t;

constexpr u32 NONE = 0xFFFFFFFFu;

// ---------------------------------------------------------------------------------------------
// Byte sink.  core/bit_coder.rs:7-48 (ByteWriter + impl for Vec<u8>): everything little-endian.
// ---------------------------------------------------------------------------------------------
struct Bytes : std::vector<u8> {
  void w8(u8 v) { push_back(v); }
  void w16(u16 v) { w8((u8)v); w8((u8)(v >> 8)); }
  void w24(u32 v) { w8((u8)v); w8((u8)(v >> 8)); w8((u8)(v >> 16)); }
  void w32(u32 v) { w16((u16)v); w16((u16)(v >> 16)); }
  void wf32(float f) { u32 b; std::memcpy(&b, &f, 4); w32(b); }   // core/shared.rs:447-452
  void wi32(i32 v) { w32((u32)v); }
  void append(const std::vector<u8>& o) { insert(end(), o.begin(), o.end()); }
};

// utils/bit_coder.rs:20-33
void leb128_write(u64 value, Bytes& w);

// core/bit_coder.rs:90-188, LsbFirst instantiation only (the only one the encoder uses).
struct BitWriterLsb {
  Bytes& buf;
  u8 pos = 0, cur = 0;
  explicit BitWriterLsb(Bytes& b) : buf(b) {}
  void write_bits(u8 size, u64 value);
  void finish();   // Drop impl, :181-188
};
// MsbFirst variant, needed only to replay the reference's BitWriter KATs (core/bit_coder.rs:514-627).
struct BitWriterMsb {
  Bytes& buf;
  u8 pos = 0, cur = 0;
  explicit BitWriterMsb(Bytes& b) : buf(b) {}
  void write_bits(u8 size, u64 value);
  void finish();
};

// ---------------------------------------------------------------------------------------------
// Attribute model.  core/attribute/mod.rs:26-49.
// ---------------------------------------------------------------------------------------------
enum AttType : u8 { Position = 0, Normal = 1, Color = 2, TextureCoordinate = 3, Custom = 4,
                    Tangent = 5, Material = 6, Joint = 7, Weight = 8 };   // :648-661 wire ids
enum Domain : u8 { DomPosition = 0, DomCorner = 1 };                      // :705-710
// ComponentDataType; the value is the *write* id (core/attribute/mod.rs:568-582, quirk Q14).
enum CompType : u8 { U8 = 1, I8 = 2, U16 = 3, I16 = 4, U32 = 5, I32 = 6, U64 = 7, I64 = 8, F32 = 9, F64 = 10 };
int comp_size(CompType t);

struct Attribute {
  u32 id = 0;
  AttType type = Position;
  Domain domain = DomPosition;
  CompType ctype = F32;
  int ncomp = 3;
  std::vector<u8> data;          // unique values, AoS, native byte order (LE)
  bool has_map = false;          // point_to_att_val_map: Option<..>
  std::vector<u32> p2v;
  std::vector<u32> parents;      // AttributeId list

  size_t value_size() const { return (size_t)comp_size(ctype) * ncomp; }
  size_t num_unique() const { return value_size() ? data.size() / value_size() : 0; }
  size_t len() const { return has_map ? p2v.size() : num_unique(); }      // :216-223
  u32 val_idx(u32 p) const;                                                // :230-246 (asserts)
  const float* f32_at(u32 vi) const { return reinterpret_cast<const float*>(data.data()) + (size_t)vi * ncomp; }
  const i32* i32_at(u32 vi) const { return reinterpret_cast<const i32*>(data.data()) + (size_t)vi * ncomp; }
};

struct Mesh {
  std::vector<std::array<u32, 3>> faces;   // point indices
  std::vector<Attribute> atts;
};

struct Options {
  bool faithful = false;   // true: keep the reference's O(V^2) scans literally (small meshes only)
  int pos_bits = 11;       // portabilization/mod.rs:118-121 (default()); parametrised for BASELINE config 5
  int uv_bits = 10;        // :131-134
  int generic_bits = 11;   // :141 fallthrough
  bool positions_delta = false;  // internal-config variant of BASELINE config 2: DeltaPrediction + Difference
};

// core/attribute/mod.rs:394-452  (Attribute::from → remove_duplicate_values)
void remove_duplicate_values(Attribute& a, bool faithful);
// core/mesh/builder.rs:62-90
std::string mesh_build(std::vector<Attribute> atts, std::vector<std::array<u32, 3>> faces, bool faithful, Mesh& out);
// io/obj/mod.rs:14-42 (+ tobj 4.0.3 single_index/triangulate behaviour)
std::string load_obj(const std::string& path, bool faithful, Mesh& out);

// ---------------------------------------------------------------------------------------------
// Corner tables.  core/corner_table/{mod,attribute_corner_table,all_inclusive_corner_table}.rs
// ---------------------------------------------------------------------------------------------
struct CornerTable {
  const std::vector<std::array<u32, 3>>* mesh_faces = nullptr;
  std::vector<std::array<u32, 3>> conn_faces;
  std::vector<u32> opposite_corners;
  std::vector<u32> left_most_corners;
  std::vector<u32> c2v_override;            // BTreeMap<CornerIdx,VertexIdx> as flat array, NONE = absent
  std::vector<u32> non_manifold_vertex_parents;
  u32 ncorners = 0, nverts = 0;

  std::string build(const std::vector<std::array<u32, 3>>& faces, const Attribute& pos);   // mod.rs:84-118
  u32 num_faces() const { return (u32)mesh_faces->size(); }
  u32 num_corners() const { return ncorners; }
  u32 num_vertices() const { return nverts; }
  u32 face_of(u32 c) const { return c / 3; }
  u32 point_idx(u32 c) const { return (*mesh_faces)[c / 3][c % 3]; }                       // :484-487
  u32 vertex_idx(u32 c) const {                                                             // :443-459
    if (c < c2v_override.size() && c2v_override[c] != NONE) return c2v_override[c];
    return conn_faces[c / 3][c % 3];
  }
  u32 opposite(u32 c) const { return opposite_corners[c]; }
  static u32 next(u32 c) { return c % 3 == 2 ? c - 2 : c + 1; }                             // :516-525
  static u32 previous(u32 c) { return c % 3 == 0 ? c + 2 : c - 1; }                         // :504-513
  u32 left_most_corner(u32 v) const { return left_most_corners[v]; }
  u32 swing_right(u32 c) const { u32 o = opposite(previous(c)); return o == NONE ? NONE : previous(o); }  // :20-26
  u32 swing_left(u32 c) const { u32 o = opposite(next(c)); return o == NONE ? NONE : next(o); }           // :28-34
  u32 get_left_corner(u32 c) const { return opposite(previous(c)); }                        // :40-42
  u32 get_right_corner(u32 c) const { return opposite(next(c)); }                           // :44-46
  bool is_on_boundary(u32 v) const { return swing_left(left_most_corner(v)) == NONE; }      // :36-38

 private:
  void compute_table();                 // :252-340
  bool contains_non_manifold_edges();   // :121-145
  void handle_non_manifold_edges();     // :149-234
  void compute_left_most_corners();     // :342-416
};

struct AttributeCornerTable {           // attribute_corner_table.rs:6-13
  std::vector<u32> corner_to_vertex;
  std::vector<u32> vertex_to_attribute_map;
  std::vector<u8> is_edge_on_seam;
  std::vector<u8> is_vertex_on_seam;
  std::vector<u32> left_most_corners;
  u32 nverts = 0;
  void build(const CornerTable& ct, const Attribute& att);                                  // :16-77
  void recompute_vertices(const Attribute& att, const CornerTable& ct);                     // :79-137
  u32 opposite(u32 c, const CornerTable& ct) const { return is_edge_on_seam[c] ? NONE : ct.opposite(c); }  // :160-166
  u32 swing_left(u32 c, const CornerTable& ct) const { u32 o = opposite(CornerTable::next(c), ct); return o == NONE ? NONE : CornerTable::next(o); }
  u32 swing_right(u32 c, const CornerTable& ct) const { u32 o = opposite(CornerTable::previous(c), ct); return o == NONE ? NONE : CornerTable::previous(o); }
};

// A uniform view used by the sequencer and the predictors: either the universal table or
// RefAttributeCornerTable (all_inclusive_corner_table.rs:70-109).
struct TableView {
  const CornerTable* ct = nullptr;
  const AttributeCornerTable* at = nullptr;   // null → universal
  u32 num_faces() const { return ct->num_faces(); }
  u32 num_corners() const { return ct->num_corners(); }
  u32 num_vertices() const { return at ? at->nverts : ct->num_vertices(); }
  u32 face_of(u32 c) const { return c / 3; }
  u32 point_idx(u32 c) const { return ct->point_idx(c); }                                   // :88-90
  u32 vertex_idx(u32 c) const { return at ? at->corner_to_vertex[c] : ct->vertex_idx(c); }
  u32 opposite(u32 c) const { return at ? at->opposite(c, *ct) : ct->opposite(c); }
  static u32 next(u32 c) { return CornerTable::next(c); }
  static u32 previous(u32 c) { return CornerTable::previous(c); }
  u32 left_most_corner(u32 v) const { return at ? at->left_most_corners[v] : ct->left_most_corner(v); }
  u32 swing_right(u32 c) const { u32 o = opposite(previous(c)); return o == NONE ? NONE : previous(o); }
  u32 swing_left(u32 c) const { u32 o = opposite(next(c)); return o == NONE ? NONE : next(o); }
  u32 get_left_corner(u32 c) const { return opposite(previous(c)); }
  u32 get_right_corner(u32 c) const { return opposite(next(c)); }
  bool is_on_boundary(u32 v) const { return swing_left(left_most_corner(v)) == NONE; }
};

// ---------------------------------------------------------------------------------------------
// Connectivity (Edgebreaker).  encode/connectivity/edgebreaker.rs
// ---------------------------------------------------------------------------------------------
struct ConnOutput {                      // edgebreaker.rs:98-101
  CornerTable ct;
  std::vector<AttributeCornerTable> att_tables;
  std::vector<u32> corners_of_edgebreaker;
  // diagnostics for tests
  std::string symbols;                   // CLERS string in traversal order
};
std::string encode_connectivity(const Mesh& mesh, Bytes& w, ConnOutput& out);              // :128-193, :458-530

// shared/attribute/sequence.rs:48-151
std::vector<u32> compute_sequence(const TableView& tv, std::vector<u32> seeds, bool faithful);

// ---------------------------------------------------------------------------------------------
// Entropy.  encode/entropy/{rans,symbol_coding}.rs, shared/entropy/mod.rs
// ---------------------------------------------------------------------------------------------
struct RansCoder {                       // rans.rs:10-69
  u32 precision;
  u64 state, l_base;
  std::vector<u32> freq, cum;
  Bytes out;
  std::string init(const std::vector<u64>& dist, u32 precision_bits);   // RansCoder::new + rans_build_tables
  std::string write(u64 idx);
  std::string flush(Bytes& dst);
};
struct RabsCoder {                       // rans.rs:71-128  (precision 8, L = 4096)
  u64 state = 4096, p0 = 0;
  Bytes out;
  explicit RabsCoder(u64 zero_prob) : p0(zero_prob) {}
  void write(u8 bit);
  std::string flush(Bytes& dst);
};
// symbol_coding.rs:17-55 (DirectCoded only is reachable: attribute_encoder.rs:351)
std::string encode_symbols_direct(const std::vector<u32>& symbols, Bytes& w);
// (SymbolEncodingMethod::LengthCoded, symbol_coding.rs:25-42,69-107, is unreachable from encode():
//  attribute_encoder.rs:351 always passes DirectCoded — not restated.)
// Inverse specs (decode/entropy/rans.rs:58-69,106-127,139-200) used only for round-trip self-checks.
std::string rans_decode_stream(const u8* data, size_t len, const std::vector<u64>& dist, u32 precision_bits, size_t n, std::vector<u32>& out);
std::string rabs_decode_stream(const u8* data, size_t len, u64 zero_prob, size_t n, std::vector<u8>& out);
std::string decode_symbols_direct(const u8* data, size_t len, size_t n, std::vector<u32>& out, size_t* consumed);

// ---------------------------------------------------------------------------------------------
// Attribute section + whole file.
// ---------------------------------------------------------------------------------------------
using Blobs = std::map<std::string, std::vector<u8>>;   // named debug dumps for the parity tests
std::string encode_attributes(const std::vector<Attribute>& atts, const ConnOutput& conn, const Options& opt, Bytes& w, Blobs* dump);  // encode/attribute/mod.rs:13-93
std::string encode_mesh(const Mesh& mesh, const Options& opt, Bytes& w, Blobs* dump);   // encode/mod.rs:59-97

// ---------------------------------------------------------------------------------------------
// The attribute section read backwards (orc_decode.cpp): what a decoder recovers once the connectivity stage is known.
// ---------------------------------------------------------------------------------------------
struct DecodedAttribute {
  u32 id = 0;
  AttType type = Position;
  CompType ctype = F32;
  int ncomp = 0, ncomp_port = 0;        // components of the attribute / of its portable (quantized) form
  u8 domain = 0, port = 0, scheme = 0, transform = 0;
  std::vector<u32> seq;                 // the attribute's coding sequence (corners)
  std::vector<i32> portable;            // quantized values, sequence order, ncomp_port per entry
  std::vector<float> values;            // dequantized values, sequence order, ncomp per entry (ToBits: the raw 4-byte values)
  std::vector<i32> by_vertex;           // (Position) quantized values by universal vertex
};
std::string decode_attributes(const u8* data, size_t len, const ConnOutput& conn, std::vector<DecodedAttribute>& out, size_t* consumed);
void oct_orthogonal_inverse(const i32* pred, const i32* corr, i32* orig);   // oct_orthogonal.rs:23-74 inverted
void oct_orthogonal_map(const i32* orig, const i32* pred, i32* corr);       // oct_orthogonal.rs:23-74

// wall-clock seconds of the last encode_mesh (BASELINE.md §3 stage split): [0] connectivity, [1] attribute section, [2] sequencer part
// of [1], [3] corner tables (part of [0]; Edgebreaker + connectivity bytes = [0] - [3]), [4] quantize (portabilization), [5] predict,
// [6] prediction transform, [7] histogram + table normalisation/serialisation, [8] rANS + rABS coders, [9] symbols through the rANS coder
constexpr int kStageSlots = 10;
extern double g_stage_seconds[kStageSlots];
struct StageTimer {   // adds the scope's wall time to g_stage_seconds[slot]
  int slot;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit StageTimer(int s) : slot(s) {}
  ~StageTimer() { g_stage_seconds[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
extern double g_rans_symbols;   // (slot 9's unit)

// helpers shared by .cpp files
i32 to_positive_i32(i32 v);   // utils/mod.rs:152-158
template <class T> inline void blob_put(Blobs* b, const std::string& k, const std::vector<T>& v) {
  if (!b) return;
  std::vector<u8> raw(v.size() * sizeof(T));
  if (!v.empty()) std::memcpy(raw.data(), v.data(), raw.size());
  (*b)[k] = std::move(raw);
}

}  // namespace orc
