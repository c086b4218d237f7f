// oracle/orc_capi.cpp — TEST INFRASTRUCTURE (see oracle.hpp header).
// Flat C entry points so tests/ (ctypes) and bench.py's cpu_baseline leg can drive the restatement.
#include <chrono>
#include <cstdio>
#include <memory>

#include "oracle.hpp"

using namespace orc;

namespace {
struct Session {
  Mesh mesh;
  Bytes drc;
  Blobs blobs;
  std::string err;
  double last_encode_seconds = 0.0;
  std::vector<DecodedAttribute> decoded;
  std::vector<std::vector<u32>> decoded_points;
};
thread_local std::string g_err;
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_session_new() { return new Session(); }
void orc_session_free(void* s) { delete static_cast<Session*>(s); }

// Load an OBJ through the restated tobj + MeshBuilder path.
int orc_load_obj(void* sp, const char* path, int faithful) {
  auto* s = static_cast<Session*>(sp);
  g_err = load_obj(path, faithful != 0, s->mesh);
  return g_err.empty() ? 0 : 1;
}

// Add an attribute from raw values (one row per point, `count` rows) — goes through
// Attribute::from (value dedup) like MeshBuilder::add_attribute.  Call orc_build afterwards.
struct PendingAtt { Attribute a; };
static thread_local std::vector<Attribute> g_pending;
static thread_local std::vector<std::array<u32, 3>> g_pending_faces;

void orc_builder_reset() { g_pending.clear(); g_pending_faces.clear(); }
int orc_builder_add_attribute(const void* data, uint32_t count, int att_type, int domain, int comp_type, int ncomp,
                              const uint32_t* parents, uint32_t nparents, int faithful) {
  Attribute a;
  a.id = (u32)g_pending.size();
  a.type = (AttType)att_type; a.domain = (Domain)domain; a.ctype = (CompType)comp_type; a.ncomp = ncomp;
  a.data.resize((size_t)count * a.value_size());
  if (count) std::memcpy(a.data.data(), data, a.data.size());
  a.parents.assign(parents, parents + nparents);
  remove_duplicate_values(a, faithful != 0);
  g_pending.push_back(std::move(a));
  return (int)g_pending.size() - 1;
}
void orc_builder_set_faces(const uint32_t* idx, uint32_t nfaces) {
  g_pending_faces.resize(nfaces);
  for (u32 f = 0; f < nfaces; ++f) g_pending_faces[f] = {idx[3 * f], idx[3 * f + 1], idx[3 * f + 2]};
}
int orc_build(void* sp, int faithful) {
  auto* s = static_cast<Session*>(sp);
  g_err = mesh_build(std::move(g_pending), std::move(g_pending_faces), faithful != 0, s->mesh);
  g_pending.clear(); g_pending_faces.clear();
  return g_err.empty() ? 0 : 1;
}

// Mesh accessors
uint32_t orc_num_faces(void* sp) { return (u32) static_cast<Session*>(sp)->mesh.faces.size(); }
const uint32_t* orc_faces(void* sp) { auto& f = static_cast<Session*>(sp)->mesh.faces; return f.empty() ? nullptr : f[0].data(); }
uint32_t orc_num_attributes(void* sp) { return (u32) static_cast<Session*>(sp)->mesh.atts.size(); }
// info[8] = {id, type, domain, comp_type, ncomp, num_unique, len, has_map}
void orc_attribute_info(void* sp, uint32_t i, uint32_t* info) {
  auto& a = static_cast<Session*>(sp)->mesh.atts[i];
  info[0] = a.id; info[1] = a.type; info[2] = a.domain; info[3] = a.ctype; info[4] = (u32)a.ncomp;
  info[5] = (u32)a.num_unique(); info[6] = (u32)a.len(); info[7] = a.has_map ? 1 : 0;
}
const void* orc_attribute_data(void* sp, uint32_t i) { return static_cast<Session*>(sp)->mesh.atts[i].data.data(); }
const uint32_t* orc_attribute_map(void* sp, uint32_t i) { auto& a = static_cast<Session*>(sp)->mesh.atts[i]; return a.has_map ? a.p2v.data() : nullptr; }
uint32_t orc_attribute_num_parents(void* sp, uint32_t i) { return (u32) static_cast<Session*>(sp)->mesh.atts[i].parents.size(); }
const uint32_t* orc_attribute_parents(void* sp, uint32_t i) { return static_cast<Session*>(sp)->mesh.atts[i].parents.data(); }

// Encode the session's mesh.  opts[5] = {faithful, pos_bits, uv_bits, generic_bits, positions_delta}
int orc_encode(void* sp, const int* opts, int want_dump) {
  auto* s = static_cast<Session*>(sp);
  Options o;
  if (opts) { o.faithful = opts[0] != 0; o.pos_bits = opts[1]; o.uv_bits = opts[2]; o.generic_bits = opts[3]; o.positions_delta = opts[4] != 0; }
  s->drc.clear();
  s->blobs.clear();
  auto t0 = std::chrono::steady_clock::now();
  g_err = encode_mesh(s->mesh, o, s->drc, want_dump ? &s->blobs : nullptr);
  s->last_encode_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return g_err.empty() ? 0 : 1;
}
void orc_last_stage_seconds(double* out) { out[0] = g_stage_seconds[0]; out[1] = g_stage_seconds[1]; out[2] = g_stage_seconds[2]; }
// all kStageSlots slots of oracle.hpp + the number of symbols slot 9 coded (out: kStageSlots + 1 doubles)
void orc_last_stage_split(double* out) { for (int k = 0; k < kStageSlots; ++k) out[k] = g_stage_seconds[k]; out[kStageSlots] = g_rans_symbols; }
double orc_last_encode_seconds(void* sp) { return static_cast<Session*>(sp)->last_encode_seconds; }
const uint8_t* orc_drc(void* sp, uint64_t* len) { auto* s = static_cast<Session*>(sp); *len = s->drc.size(); return s->drc.data(); }
const uint8_t* orc_blob(void* sp, const char* key, uint64_t* len) {
  auto* s = static_cast<Session*>(sp);
  auto it = s->blobs.find(key);
  if (it == s->blobs.end()) { *len = 0; return nullptr; }
  *len = it->second.size();
  return it->second.data();
}

// ---- the attribute section read backwards (orc_decode.cpp) ----
// Decodes `data` (an attribute section: the oracle's own, or the product's) against the connectivity stage of the session's mesh.
// Results stay in the session: orc_decoded_count / orc_decoded_info / orc_decoded_array.
int orc_decode_attributes(void* sp, const uint8_t* data, uint64_t len, uint64_t* consumed) {
  auto* s = static_cast<Session*>(sp);
  Bytes scratch;
  ConnOutput conn;
  g_err = encode_connectivity(s->mesh, scratch, conn);   // what a decoder knows after ITS connectivity stage: the same tables and seeds
  if (!g_err.empty()) return 1;
  size_t used = 0;
  s->decoded.clear();
  g_err = decode_attributes(data, (size_t)len, conn, s->decoded, &used);
  if (consumed) *consumed = used;
  if (g_err.empty()) {   // the points behind the sequence entries, for the caller's comparisons
    s->decoded_points.assign(s->decoded.size(), {});
    for (size_t i = 0; i < s->decoded.size(); ++i) for (u32 c : s->decoded[i].seq) s->decoded_points[i].push_back(conn.ct.point_idx(c));
  }
  return g_err.empty() ? 0 : 1;
}
uint32_t orc_decoded_count(void* sp) { return (u32) static_cast<Session*>(sp)->decoded.size(); }
// info[8] = {id, type, ncomp, ncomp_port, port, scheme, transform, entries}
void orc_decoded_info(void* sp, uint32_t i, uint32_t* info) {
  auto& a = static_cast<Session*>(sp)->decoded[i];
  info[0] = a.id; info[1] = a.type; info[2] = (u32)a.ncomp; info[3] = (u32)a.ncomp_port; info[4] = a.port; info[5] = a.scheme; info[6] = a.transform; info[7] = (u32)a.seq.size();
}
// which: 0 portable (i32), 1 values (f32 / raw 4-byte), 2 point index of every sequence entry (u32)
const void* orc_decoded_array(void* sp, uint32_t i, int which, uint64_t* count) {
  auto* s = static_cast<Session*>(sp);
  auto& a = s->decoded[i];
  if (which == 0) { *count = a.portable.size(); return a.portable.data(); }
  if (which == 1) { *count = a.values.size(); return a.values.data(); }
  *count = s->decoded_points[i].size();
  return s->decoded_points[i].data();
}
// oct_orthogonal.rs:23-74 and its inverse on one pair (KAT: the map must be invertible on the whole octahedral grid)
void orc_oct_orthogonal(const int32_t* orig, const int32_t* pred, int32_t* corr, int32_t* back) { oct_orthogonal_map(orig, pred, corr); oct_orthogonal_inverse(pred, corr, back); }

// ---- small KAT hooks ----
uint64_t orc_leb128(uint64_t v, uint8_t* out) { Bytes b; leb128_write(v, b); std::memcpy(out, b.data(), b.size()); return b.size(); }
// ops: array of (size,value) pairs; msb=1 → MsbFirst
uint64_t orc_bitwriter(const uint64_t* ops, uint32_t nops, int msb, uint8_t* out) {
  Bytes b;
  if (msb) { BitWriterMsb w(b); for (u32 i = 0; i < nops; ++i) w.write_bits((u8)ops[2 * i], ops[2 * i + 1]); w.finish(); }
  else { BitWriterLsb w(b); for (u32 i = 0; i < nops; ++i) w.write_bits((u8)ops[2 * i], ops[2 * i + 1]); w.finish(); }
  std::memcpy(out, b.data(), b.size());
  return b.size();
}
// raw RansCoder with a given (already normalised) table: returns byte count, -1 on error
int64_t orc_rans_encode_raw(const uint64_t* dist, uint32_t ndist, uint32_t precision, const uint32_t* syms, uint64_t n, uint8_t* out, uint64_t cap) {
  RansCoder rc;
  std::vector<u64> d(dist, dist + ndist);
  g_err = rc.init(d, precision);
  if (!g_err.empty()) return -1;
  for (u64 i = 0; i < n; ++i) { g_err = rc.write(syms[i]); if (!g_err.empty()) return -1; }
  Bytes b;
  g_err = rc.flush(b);
  if (!g_err.empty() || b.size() > cap) return -1;
  std::memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
}
int orc_rans_decode_raw(const uint8_t* data, uint64_t len, const uint64_t* dist, uint32_t ndist, uint32_t precision, uint64_t n, uint32_t* out) {
  std::vector<u64> d(dist, dist + ndist);
  std::vector<u32> o;
  g_err = rans_decode_stream(data, len, d, precision, n, o);
  if (!g_err.empty()) return 1;
  std::memcpy(out, o.data(), o.size() * 4);
  return 0;
}
int64_t orc_rabs_encode(uint32_t zero_prob, const uint8_t* bits, uint64_t n, uint8_t* out, uint64_t cap) {
  RabsCoder rc(zero_prob);
  for (u64 i = 0; i < n; ++i) rc.write(bits[i]);
  Bytes b;
  g_err = rc.flush(b);
  if (!g_err.empty() || b.size() > cap) return -1;
  std::memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
}
int orc_rabs_decode(const uint8_t* data, uint64_t len, uint32_t zero_prob, uint64_t n, uint8_t* out) {
  std::vector<u8> o;
  g_err = rabs_decode_stream(data, len, zero_prob, n, o);
  if (!g_err.empty()) return 1;
  std::memcpy(out, o.data(), o.size());
  return 0;
}
int64_t orc_encode_symbols(const uint32_t* syms, uint64_t n, uint8_t* out, uint64_t cap) {
  std::vector<u32> s(syms, syms + n);
  Bytes b;
  g_err = encode_symbols_direct(s, b);
  if (!g_err.empty() || b.size() > cap) return -1;
  std::memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
}
int orc_decode_symbols(const uint8_t* data, uint64_t len, uint64_t n, uint32_t* out, uint64_t* consumed) {
  std::vector<u32> o;
  size_t used = 0;
  g_err = decode_symbols_direct(data, len, n, o, &used);
  if (!g_err.empty()) return 1;
  std::memcpy(out, o.data(), o.size() * 4);
  *consumed = used;
  return 0;
}

}  // extern "C"
