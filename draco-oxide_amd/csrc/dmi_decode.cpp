// dmi_decode.cpp — dmi_decode_attributes: the attribute section read back ("decoder-side inverse kernels").
//
// What a decoder does once its connectivity stage has rebuilt the corner tables (here: handed over by the caller, exactly the arrays
// dmi_encode_attributes takes), split by where each stage can run:
//   host cores (serial, like their encoders)   frequency table + rANS symbol decoding (decode/entropy/symbol_coding.rs:125-210,
//                                              rans.rs:58-69), rABS metadata bits (rans.rs:106-127), and the predictions whose
//                                              inputs are values of the SAME attribute decoded earlier: parallelogram + wrapped
//                                              difference, delta + difference, texture coordinates (each entry needs its
//                                              predecessors: a dependency chain through the whole sequence)
//   device (data-parallel)                     normals — their prediction only reads the already decoded POSITIONS — with the
//                                              oct-orthogonal transform inverted (k_decode_normals), and the dequantization of every
//                                              attribute scattered to its points (k_dequantize)
// The layout parsed here is the one encode_attributes writes (encode/attribute/mod.rs:26-57, attribute_encoder.rs:159-160,344-386);
// the reference's own decode/attribute/* is an unbuilt prototype of an older layout, so everything above the entropy layer is the
// encoder spec inverted (the same derivation as oracle/orc_decode.cpp, which the tests compare this against).
// No CPU fallback: without a HIP device the call returns DMI_ERR_NO_DEVICE.
#include <algorithm>
#include <cmath>
#include <limits>
#include <chrono>
#include <memory>
#include <thread>

#include "dmi_device.hpp"
#include "dmi_host.hpp"
#include "host_chains.hpp"

namespace dmi {
thread_local dmi_decode_timings g_last_decode{};   // dmi_last_decode_timings
thread_local bool g_inside_decode_mesh = false;     // dmi_decode_mesh is the caller: its connectivity / table times stay
namespace {

#define HIP_TRY_D(expr)                                                                                            \
  do {                                                                                                             \
    hipError_t e_ = (expr);                                                                                        \
    if (e_ != hipSuccess) {                                                                                        \
      const bool nodev = (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorInsufficientDriver || e_ == hipErrorNotInitialized); \
      return host_fail(nodev ? DMI_ERR_NO_DEVICE : (e_ == hipErrorOutOfMemory ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_HIP), std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                                                              \
  } while (0)

inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int64_t wadd64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
inline int64_t wsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
inline int64_t wmul64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
inline int64_t wabs64(int64_t a) { return a < 0 ? (int64_t)(0 - (uint64_t)a) : a; }
inline int64_t wdiv64(int64_t a, int64_t b) { if (a == std::numeric_limits<int64_t>::min() && b == -1) return a; return a / b; }
inline int32_t from_positive(uint32_t s) { return (s & 1u) ? (int32_t)(0u - ((s >> 1) + 1u)) : (int32_t)(s >> 1); }   // utils/mod.rs:152-158 inverted

uint64_t int_sqrt(uint64_t value) {   // mesh_prediction_for_texture_coordinates.rs:32-48
  if (value == 0) return 0;
  uint64_t act = value, sq = 1;
  while (act >= 2) { sq *= 2; act /= 4; }
  sq = (sq + value / sq) / 2;
  while (sq * sq > value) sq = (sq + value / sq) / 2;
  return sq;
}

struct Reader {
  const uint8_t* p; size_t n, at = 0;
  bool ok = true;
  uint8_t r8() { if (at >= n) { ok = false; return 0; } return p[at++]; }
  uint32_t r32() { uint32_t v = 0; for (int k = 0; k < 4; ++k) v |= (uint32_t)r8() << (8 * k); return v; }
  float rf32() { const uint32_t b = r32(); float f; std::memcpy(&f, &b, 4); return f; }
  // at most ten bytes; a tenth byte may only carry bit 63 (anything longer or larger is a malformed section, not a wrapped shift)
  uint64_t leb() {
    uint64_t v = 0; uint32_t sh = 0; uint8_t b;
    do {
      b = r8();
      if (sh == 63 && (b & 0x7E)) { ok = false; return 0; }
      v |= (uint64_t)(b & 0x7F) << sh; sh += 7;
    } while ((b & 0x80) && ok && sh < 70);
    if (b & 0x80) ok = false;
    return v;
  }
};

// What the parse pass records per attribute, and what its entropy thread leaves behind
struct AttPlan {
  int N = 0;
  std::vector<uint32_t> seq_own; const uint32_t* seq = nullptr; uint32_t n = 0;
  std::vector<uint32_t> freq; uint32_t P = 0; const uint8_t* rans = nullptr; size_t rans_bytes = 0;
  bool has_rabs = false; uint8_t zp = 0; const uint8_t* rabs = nullptr; size_t rabs_bytes = 0; uint64_t rabs_count = 0;
  int32_t t_min = 0, t_max = 0;
  float q_min[4] = {0, 0, 0, 0}, q_range = 0; int q_bits = 0;
  std::vector<uint32_t> sym; std::vector<uint8_t> bits;
  int rc = 0; std::string err;
};

// DirectCoded symbols (decode/entropy/symbol_coding.rs:125-210 + RansSymbolDecoder::new rans.rs:139-200): method, bit_length,
// frequency table with zero-run tokens, leb128 length, stream — located, not decoded (the attribute's host thread does that)
int locate_symbols(Reader& r, AttPlan& pl) {
  if (r.r8() != 1) return host_fail(DMI_ERR_ENTROPY, "symbols are not direct coded");
  const uint8_t bl = r.r8();
  if (bl < 1 || bl > 18) return host_fail(DMI_ERR_ENTROPY, "bad symbol bit length");
  static const uint8_t prec_of[19] = {0, 12, 12, 12, 12, 12, 12, 12, 12, 13, 15, 16, 18, 19, 20, 20, 20, 20, 20};
  pl.P = prec_of[bl];
  const uint64_t num_symbols = r.leb();
  if (!r.ok || num_symbols > ((uint64_t)1 << 21)) return host_fail(DMI_ERR_ENTROPY, "bad frequency table size");
  pl.freq.assign((size_t)num_symbols, 0);
  for (uint64_t i = 0; i < num_symbols; ++i) {
    const uint8_t b = r.r8();
    const uint32_t token = b & 3u;
    if (token == 3) {
      const uint32_t offset = b >> 2;
      if (i + offset >= num_symbols) return host_fail(DMI_ERR_ENTROPY, "zero run past the frequency table");
      i += offset;
    } else {
      uint32_t count = b >> 2;
      for (uint32_t j = 0; j < token; ++j) count |= (uint32_t)r.r8() << (8 * (j + 1) - 2);
      pl.freq[(size_t)i] = count;
    }
  }
  const uint64_t nbytes = r.leb();
  if (!r.ok || nbytes > r.n - r.at) return host_fail(DMI_ERR_ENTROPY, "symbol stream past the section");   // (no addition: nbytes is attacker-controlled)
  pl.rans = r.p + r.at; pl.rans_bytes = (size_t)nbytes;
  r.at += (size_t)nbytes;
  return DMI_OK;
}

struct DevBuf {   // (a decode call owns a handful of device arrays)
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int upload(const void* src, size_t bytes, hipStream_t s) {
    HIP_TRY_D(hipMalloc(&p, bytes ? bytes : 4));
    if (bytes) HIP_TRY_D(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s));
    return DMI_OK;
  }
  int alloc(size_t bytes) { HIP_TRY_D(hipMalloc(&p, bytes ? bytes : 4)); return DMI_OK; }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

struct Owner {
  std::vector<dmi_decoded_attribute> atts;   // (values pointers are filled in as the attributes finish)
  std::vector<std::vector<float>> values;
};

}  // namespace
}  // namespace dmi

using namespace dmi;

extern "C" {

int dmi_last_decode_timings(dmi_decode_timings* t) {
  if (!t) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null argument");
  *t = g_last_decode;
  return DMI_OK;
}

void dmi_decoded_free(dmi_decoded* d) {
  if (!d) return;
  delete static_cast<Owner*>(d->owner);
  d->owner = nullptr; d->attributes = nullptr; d->num_attributes = 0;
}

static int decode_attributes_impl(const uint8_t* section, size_t len, const dmi_corner_table* tables, uint32_t n_tables, const uint32_t* seeds, uint32_t n_seeds,
                                  uint32_t num_points, const dmi_config* cfg_in, dmi_decoded* out);
// (no exception crosses the C boundary: a damaged file that asks for more memory than there is comes back as an error code)
int dmi_decode_attributes(const uint8_t* section, size_t len, const dmi_corner_table* tables, uint32_t n_tables, const uint32_t* seeds, uint32_t n_seeds,
                          uint32_t num_points, const dmi_config* cfg_in, dmi_decoded* out) {
  try {
    return decode_attributes_impl(section, len, tables, n_tables, seeds, n_seeds, num_points, cfg_in, out);
  } catch (const std::bad_alloc&) { return host_fail(DMI_ERR_OUT_OF_MEMORY, "out of memory decoding the attribute section"); }
  catch (const std::exception& e) { return host_fail(DMI_ERR_ENTROPY, std::string("attribute decoding failed: ") + e.what()); }
}
static int decode_attributes_impl(const uint8_t* section, size_t len, const dmi_corner_table* tables, uint32_t n_tables, const uint32_t* seeds, uint32_t n_seeds,
                                  uint32_t num_points, const dmi_config* cfg_in, dmi_decoded* out) {
  if (!section || !tables || !out || n_tables == 0) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null argument");
  const auto t_call0 = std::chrono::steady_clock::now();
  { const float c = g_last_decode.connectivity_ms, tb = g_last_decode.tables_ms; g_last_decode = dmi_decode_timings{}; if (g_inside_decode_mesh) { g_last_decode.connectivity_ms = c; g_last_decode.tables_ms = tb; } }
  dmi_config cfg{};
  if (cfg_in) cfg = *cfg_in;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return host_fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  HIP_TRY_D(hipSetDevice(cfg.device));
  hipStream_t s = static_cast<hipStream_t>(cfg.stream);
  struct OwnStream { hipStream_t s = nullptr; ~OwnStream() { if (s) (void)hipStreamDestroy(s); } } own;
  if (!s) { HIP_TRY_D(hipStreamCreate(&own.s)); s = own.s; }

  Reader r{section, len};
  const uint32_t n_atts = r.r8();                                                  // encode/attribute/mod.rs:26
  if (n_atts == 0 || n_atts > n_tables) return host_fail(DMI_ERR_INVALID_ARGUMENT, "the section codes " + std::to_string(n_atts) + " attributes, " + std::to_string(n_tables) + " corner tables given");
  std::vector<uint8_t> domain(n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) { r.r8(); domain[i] = r.r8(); r.r8(); }   // :30-39 (data id, domain, traversal)
  std::unique_ptr<Owner> owner(new Owner());
  owner->atts.resize(n_atts);
  owner->values.resize(n_atts);
  std::vector<uint8_t> port(n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) {                                         // :43-57
    dmi_decoded_attribute& a = owner->atts[i];
    if (r.r8() != 1) return host_fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "attribute decoder with more than one attribute");
    a.att_type = r.r8(); a.component_type = r.r8(); a.num_components = r.r8();
    r.r8();                                                                        // normalized
    a.unique_id = r.r8();
    a.portabilization = port[i] = r.r8();
    a.domain = domain[i];
    a.num_points = num_points;
    if (a.num_components < 1 || a.num_components > 4) return host_fail(DMI_ERR_UNSUPPORTED_NUM_COMPONENTS, "components must be 1..4");
  }
  if (!r.ok) return host_fail(DMI_ERR_ENTROPY, "truncated attribute headers");
  const uint32_t F = tables[0].num_faces;
  const size_t C = (size_t)F * 3;
  for (uint32_t i = 0; i < n_atts; ++i) {
    if (tables[i].num_faces != F || !tables[i].corner_to_point || !tables[i].corner_to_vertex || !tables[i].opposite) return host_fail(DMI_ERR_INVALID_ARGUMENT, "corner table arrays missing");
    for (size_t c = 0; c < C; ++c) {
      if (tables[i].corner_to_vertex[c] >= tables[i].num_vertices || tables[i].corner_to_point[c] >= num_points || (tables[i].opposite[c] != kNone && tables[i].opposite[c] >= C))
        return host_fail(DMI_ERR_INVALID_ARGUMENT, "corner table " + std::to_string(i) + ": entry out of range");
    }
  }
  // the corner tables on the device (shared by the kernels of every attribute)
  DevBuf d_c2p, d_last;
  int rc = d_c2p.upload(tables[0].corner_to_point, C * 4, s);
  if (rc) return rc;
  if ((rc = d_last.alloc((size_t)num_points * 4))) return rc;
  HIP_TRY_D(hipMemsetAsync(d_last.p, 0, (size_t)num_points * 4, s));
  launch_last_corners(d_c2p.as<uint32_t>(), C, d_last.as<uint32_t>(), s);
  std::vector<int32_t> pos_by_vertex;   // decoded positions by UNIVERSAL vertex (the parent of normals and texture coordinates)
  int pos_att = -1;
  DevBuf d_pos, d_c2v_pos;

  // ---- 1. parse: every length in the section is explicit, so the blocks of all attributes are located without decoding anything ----
  std::vector<AttPlan> plans(n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) {
    dmi_decoded_attribute& a = owner->atts[i];
    AttPlan& pl = plans[i];
    a.scheme = r.r8(); a.transform = r.r8();                                       // attribute_encoder.rs:159-160
    if (r.r8() != 1) return host_fail(DMI_ERR_ENTROPY, "rans_encoding flag not set");
    pl.N = port[i] == 3 ? 2 : a.num_components;
    if ((rc = locate_symbols(r, pl))) return rc;
    bool meta_ok = true;
    auto read_transform_info = [&] {
      if (a.transform == 1) { pl.t_min = (int32_t)r.r32(); pl.t_max = (int32_t)r.r32(); }
      else if (a.transform == 3) { if (r.r32() != 255 || r.r32() != 127) meta_ok = false; }
    };
    auto locate_rabs = [&]() -> int {
      pl.zp = r.r8();
      const uint64_t nbytes = r.leb();
      if (!r.ok || pl.zp == 0 || nbytes > len - r.at) return host_fail(DMI_ERR_ENTROPY, "truncated rABS block");
      pl.rabs = section + r.at; pl.rabs_bytes = (size_t)nbytes; pl.has_rabs = true;
      r.at += (size_t)nbytes;
      return DMI_OK;
    };
    if (a.scheme == 6) { read_transform_info(); if ((rc = locate_rabs())) return rc; }
    else if (a.scheme == 5) { pl.rabs_count = r.r32(); if ((rc = locate_rabs())) return rc; read_transform_info(); }
    else read_transform_info();
    if (port[i] == 2) { for (int k = 0; k < a.num_components; ++k) pl.q_min[k] = r.rf32(); pl.q_range = r.rf32(); pl.q_bits = r.r8(); if (pl.q_bits < 1 || pl.q_bits > 31) meta_ok = false; }
    else if (port[i] == 3) { if (r.r8() != 8) meta_ok = false; }
    else if (port[i] != 1) return host_fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "unknown portabilization");
    if (!r.ok || !meta_ok) return host_fail(DMI_ERR_ENTROPY, "truncated or unexpected attribute metadata");
    a.bits = (uint8_t)(port[i] == 3 ? 8 : pl.q_bits);
  }
  if (r.at != len) return host_fail(DMI_ERR_ENTROPY, std::to_string(len - r.at) + " bytes left after the last attribute");

  // ---- 2. host threads: one traversal per DISTINCT corner table (attributes without seams share the universal one), then one entropy
  //         decoder per attribute (its rANS symbols + its rABS bits), all attributes at once ----
  const auto t_seq0 = std::chrono::steady_clock::now();
  std::vector<int> seq_owner(n_atts, -1);   // the attribute whose plan holds this attribute's sequence
  for (uint32_t i = 0; i < n_atts; ++i) {
    const dmi_corner_table& t = tables[i];
    if (t.sequence) { plans[i].seq = t.sequence; plans[i].n = t.sequence_len; seq_owner[i] = (int)i; continue; }
    if (!t.left_most_corner) return host_fail(DMI_ERR_INVALID_ARGUMENT, "left_most_corner needed to compute the sequence");
    for (uint32_t j = 0; j < i && seq_owner[i] < 0; ++j)
      if (!tables[j].sequence && tables[j].corner_to_vertex == t.corner_to_vertex && tables[j].opposite == t.opposite && tables[j].left_most_corner == t.left_most_corner && tables[j].num_vertices == t.num_vertices) seq_owner[i] = seq_owner[j];
    if (seq_owner[i] < 0) seq_owner[i] = (int)i;
  }
  if (n_seeds && !seeds) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null seeds");
  for (uint32_t k = 0; k < n_seeds; ++k) if (seeds[k] >= C) return host_fail(DMI_ERR_INVALID_ARGUMENT, "seed corner out of range");
  // the entropy decoders need the entry count of their attribute before its traversal has run: a traversal visits every vertex its table's
  // corners name, so the count is the table's vertex count unless the caller's table carries unused vertices — the decoders start with that
  // guess beside the traversals, and an attribute whose traversal comes back with another count is decoded again
  auto decode_entropy = [&](uint32_t i, uint32_t n_entries) {
    AttPlan* pl = &plans[i];
    const uint8_t scheme = owner->atts[i].scheme;
    pl->rc = 0; pl->err.clear();
    pl->sym.resize((size_t)n_entries * pl->N);
    if (!host_rans_decode(pl->rans, pl->rans_bytes, pl->freq.data(), (uint32_t)pl->freq.size(), pl->P, (uint64_t)n_entries * pl->N, pl->sym.data())) { pl->rc = DMI_ERR_ENTROPY; pl->err = "truncated or inconsistent rANS stream"; return; }
    if (!pl->has_rabs) return;
    const uint64_t count = scheme == 6 ? n_entries : pl->rabs_count;
    if (count > n_entries) { pl->rc = DMI_ERR_ENTROPY; pl->err = "more orientation bits than entries"; return; }
    pl->bits.resize((size_t)count);
    if (!host_rabs_decode(pl->rabs, pl->rabs_bytes, pl->zp, count, pl->bits.data())) { pl->rc = DMI_ERR_ENTROPY; pl->err = "truncated rABS stream"; return; }
    std::reverse(pl->bits.begin(), pl->bits.end());   // the encoder pushed them first to last; an ANS decoder pops the last one first
    if (scheme == 5) {   // transitions → orientations (mesh_prediction_for_texture_coordinates.rs:241-256 inverted)
      bool last = true;
      for (size_t k = (size_t)count; k-- > 0;) { if (!pl->bits[k]) last = !last; pl->bits[k] = last ? 1 : 0; }
    }
  };
  std::vector<uint32_t> n_guess(n_atts);
  {
    // traversals and entropy decoders as items of ONE guarded pool (at most host_threads() threads; an exception in a worker — a size the file
    // asked for that cannot be allocated, no thread left — comes back as a status instead of std::terminate)
    struct Item { int kind; uint32_t i; };
    std::vector<Item> items;
    for (uint32_t i = 0; i < n_atts; ++i) {
      if (seq_owner[i] != (int)i || tables[i].sequence) continue;
      const dmi_corner_table& t = tables[i];
      for (uint32_t v = 0; v < t.num_vertices; ++v) if (t.left_most_corner[v] >= C) return host_fail(DMI_ERR_INVALID_ARGUMENT, "left_most_corner entry out of range");
      items.push_back({0, i});
    }
    const size_t n_walks = items.size();
    uint64_t asked = 0;
    for (uint32_t i = 0; i < n_atts; ++i) {
      n_guess[i] = tables[i].sequence ? tables[i].sequence_len : tables[i].num_vertices;
      if (n_guess[i] > C) return host_fail(DMI_ERR_INVALID_ARGUMENT, "corner table " + std::to_string(i) + ": more vertices than corners");
      asked += (uint64_t)n_guess[i] * plans[i].N * 4 + n_guess[i];
      items.push_back({1, i});
    }
    if (asked > decode_budget_bytes()) return host_fail(DMI_ERR_OUT_OF_MEMORY, "the section's symbol arrays exceed the decode budget (DMI_DECODE_BUDGET_MB)");
    std::atomic<size_t> walks_left{n_walks};
    std::atomic<int64_t> seq_done_ns{0};
    const int st = guarded_pool(items.size(), std::max(host_threads(), (unsigned)std::min<size_t>(items.size(), 8)), [&](size_t k) {
      const Item it = items[k];
      if (it.kind == 0) {
        const dmi_corner_table& t = tables[it.i];
        TableRef tr{F, t.num_vertices, t.corner_to_vertex, t.opposite, t.left_most_corner};
        attribute_sequence(tr, seeds, n_seeds, plans[it.i].seq_own);
        if (walks_left.fetch_sub(1) == 1) seq_done_ns.store(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_seq0).count());
      } else {
        decode_entropy(it.i, n_guess[it.i]);
      }
    });
    g_last_decode.sequence_ms = (float)(seq_done_ns.load() / 1e6);
    if (st) return host_fail(st == 1 ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_ENTROPY, st == 1 ? "out of memory decoding the attribute streams" : "attribute stream decoding failed");
  }
  for (uint32_t i = 0; i < n_atts; ++i) if (seq_owner[i] == (int)i && !tables[i].sequence) { plans[i].seq = plans[i].seq_own.data(); plans[i].n = (uint32_t)plans[i].seq_own.size(); }
  for (uint32_t i = 0; i < n_atts; ++i) {
    if (seq_owner[i] != (int)i) { plans[i].seq = plans[seq_owner[i]].seq; plans[i].n = plans[seq_owner[i]].n; }
    for (uint32_t k = 0; k < plans[i].n && seq_owner[i] == (int)i; ++k) if (plans[i].seq[k] >= C) return host_fail(DMI_ERR_INVALID_ARGUMENT, "sequence entry out of range");
    if (plans[i].n != n_guess[i]) decode_entropy(i, plans[i].n);   // (a table with vertices no corner names)
  }
  for (uint32_t i = 0; i < n_atts; ++i) if (plans[i].rc) return host_fail(plans[i].rc, "attribute " + std::to_string(i) + ": " + plans[i].err);
  const auto t_inv0 = std::chrono::steady_clock::now();
  g_last_decode.entropy_ms = std::chrono::duration<float, std::milli>(t_inv0 - t_seq0).count();   // (beside the traversals: the longer of the two)
  double inverse_ms = 0;

  // ---- 3. predictions inverted in attribute order (a normal / texture coordinate needs the decoded positions) ----
  for (uint32_t i = 0; i < n_atts; ++i) {
    dmi_decoded_attribute& a = owner->atts[i];
    const dmi_corner_table& t = tables[i];
    AttPlan& pl = plans[i];
    const uint32_t* seq = pl.seq;
    const uint32_t n = pl.n;
    const int N = pl.N;
    const std::vector<uint32_t>& sym = pl.sym;
    const std::vector<uint8_t>& bits = pl.bits;
    const int32_t t_min = pl.t_min, t_max = pl.t_max;
    const float* q_min = pl.q_min;
    const float q_range = pl.q_range;
    const int q_bits = pl.q_bits;
    if ((a.scheme == 5 || a.scheme == 6) && pos_att < 0) return host_fail(DMI_ERR_BAD_PARENT, "attribute " + std::to_string(i) + " needs a decoded Position attribute");
    const auto t_att0 = std::chrono::steady_clock::now();

    std::vector<int32_t> val((size_t)t.num_vertices * N, 0);   // quantized values by this table's vertex
    DevBuf d_val, d_c2v, d_opp, d_seq, d_sym, d_flips;
    if ((rc = d_c2v.upload(t.corner_to_vertex, C * 4, s))) return rc;
    if (a.scheme == 6) {
      // ---- normals on the device: every entry independently from the decoded positions ----
      if (a.transform != 3) return host_fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "normal prediction without the oct-orthogonal transform");
      if (N != 2) return host_fail(DMI_ERR_UNSUPPORTED_NUM_COMPONENTS, "normals must portabilize to 2 components");
      if ((rc = d_opp.upload(t.opposite, C * 4, s)) || (rc = d_seq.upload(seq, (size_t)n * 4, s)) || (rc = d_sym.upload(sym.data(), sym.size() * 4, s)) ||
          (rc = d_flips.upload(bits.data(), bits.size(), s)) || (rc = d_val.alloc(val.size() * 4)))
        return rc;
      HIP_TRY_D(hipMemsetAsync(d_val.p, 0, val.size() * 4, s));
      DecodeNormalArgs da{d_seq.as<uint32_t>(), n, 0u, d_c2v_pos.as<uint32_t>(), d_opp.as<uint32_t>(), d_c2v.as<uint32_t>(), d_pos.as<int32_t>(), d_sym.as<uint32_t>(), d_flips.as<uint8_t>(),
                          d_val.as<int32_t>()};
      launch_decode_normals(da, s);
    } else {
      // ---- sequential predictions on a host core ----
      std::vector<uint32_t> when(t.num_vertices, 0);   // sequence index + 1 of a decoded vertex
      auto have = [&](uint32_t v, size_t k) { return when[v] != 0 && when[v] - 1 < k; };
      const int32_t max_diff = a.transform == 1 ? wadd(1, wsub(t_max, t_min)) : 0;
      size_t next_bit = 0;
      // the walk's loads run ahead of it in three steps (each needs what the step before it fetched): corner rows of entry k + 32, the
      // opposite corner's vertex of entry k + 20, the values / decode marks of the three or four vertices of entry k + 10
      const uint32_t* pos_c2v = pos_att >= 0 ? tables[pos_att].corner_to_vertex : nullptr;
      auto run_ahead = [&](size_t k) {
        if (k + 32 < n) { const uint32_t c1 = seq[k + 32]; __builtin_prefetch(&t.opposite[c1]); __builtin_prefetch(&t.corner_to_vertex[3 * (c1 / 3)]); if (pos_c2v && a.scheme == 5) __builtin_prefetch(&pos_c2v[3 * (c1 / 3)]); }
        if (k + 20 < n && a.scheme == 1) { const uint32_t o = t.opposite[seq[k + 20]]; if (o != kNone) __builtin_prefetch(&t.corner_to_vertex[o]); }
        if (k + 10 < n) {
          const uint32_t c3 = seq[k + 10], f3 = 3 * (c3 / 3);
          for (uint32_t j = 0; j < 3; ++j) {
            const uint32_t v = t.corner_to_vertex[f3 + j];
            __builtin_prefetch(&when[v]); __builtin_prefetch(&val[(size_t)v * N]);
            if (pos_c2v && a.scheme == 5) __builtin_prefetch(&pos_by_vertex[(size_t)pos_c2v[f3 + j] * 3]);
          }
          if (a.scheme == 1) { const uint32_t o = t.opposite[c3]; if (o != kNone) { const uint32_t v = t.corner_to_vertex[o]; __builtin_prefetch(&when[v]); __builtin_prefetch(&val[(size_t)v * N]); } }
        }
      };
      for (size_t k = 0; k < n; ++k) {
        const uint32_t c = seq[k];
        run_ahead(k);
        int32_t pred[4] = {0, 0, 0, 0};
        auto previous_value = [&] { if (k > 0) { const int32_t* v = val.data() + (size_t)t.corner_to_vertex[seq[k - 1]] * N; for (int j = 0; j < N; ++j) pred[j] = v[j]; } };
        if (a.scheme == 1) {                                                      // mesh_parallelogram_prediction.rs:186-237
          const uint32_t opp = t.opposite[c];
          bool done = false;
          if (opp != kNone) {
            const uint32_t ov = t.corner_to_vertex[opp], nv = t.corner_to_vertex[corner_next(c)], pv = t.corner_to_vertex[corner_prev(c)];
            if (have(ov, k) && have(nv, k) && have(pv, k)) {
              for (int j = 0; j < N; ++j) pred[j] = wsub(wadd(val[(size_t)nv * N + j], val[(size_t)pv * N + j]), val[(size_t)ov * N + j]);
              done = true;
            }
          }
          if (!done) previous_value();
        } else if (a.scheme == 0) {                                               // delta_prediction.rs:56-71
          previous_value();
        } else if (a.scheme == 5) {                                               // mesh_prediction_for_texture_coordinates.rs:51-81,107-219
          if (N != 2) return host_fail(DMI_ERR_UNSUPPORTED_NUM_COMPONENTS, "texture coordinates must have 2 components");
          const uint32_t nc = corner_next(c), pc = corner_prev(c);
          const uint32_t nv = t.corner_to_vertex[nc], pv = t.corner_to_vertex[pc];
          auto pos_of = [&](uint32_t corner, int64_t* o) { const int32_t* q = pos_by_vertex.data() + (size_t)tables[pos_att].corner_to_vertex[corner] * 3; o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; };
          bool done = false;
          if (have(nv, k) && have(pv, k)) {
            const int64_t nu[2] = {val[(size_t)nv * 2], val[(size_t)nv * 2 + 1]}, pu[2] = {val[(size_t)pv * 2], val[(size_t)pv * 2 + 1]};
            if (nu[0] == pu[0] && nu[1] == pu[1]) { pred[0] = (int32_t)pu[0]; pred[1] = (int32_t)pu[1]; done = true; }
            else {
              int64_t cp[3], np[3], pp[3], pn[3], cn[3];
              pos_of(c, cp); pos_of(nc, np); pos_of(pc, pp);
              for (int j = 0; j < 3; ++j) { pn[j] = wsub64(pp[j], np[j]); cn[j] = wsub64(cp[j], np[j]); }
              const uint64_t pn2 = (uint64_t)wadd64(wadd64(wmul64(pn[0], pn[0]), wmul64(pn[1], pn[1])), wmul64(pn[2], pn[2]));
              if (pn2 != 0) {
                const int64_t cdp = wadd64(wadd64(wmul64(pn[0], cn[0]), wmul64(pn[1], cn[1])), wmul64(pn[2], cn[2]));
                const int64_t pnu[2] = {wsub64(pu[0], nu[0]), wsub64(pu[1], nu[1])};
                const int64_t I64MAX = std::numeric_limits<int64_t>::max();
                const int64_t n_uv_absmax = std::max(wabs64(nu[0]), wabs64(nu[1])), pn_uv_absmax = std::max(wabs64(pnu[0]), wabs64(pnu[1]));
                const int64_t pn_absmax = std::max(std::max(wabs64(pn[0]), wabs64(pn[1])), wabs64(pn[2]));
                if (!(n_uv_absmax > wdiv64(I64MAX, (int64_t)pn2)) && !(wabs64(cdp) > wdiv64(I64MAX, pn_uv_absmax)) && !(wabs64(cdp) > wdiv64(I64MAX, pn_absmax))) {
                  int64_t x_uv[2], cxv[3];
                  for (int j = 0; j < 2; ++j) x_uv[j] = wadd64(wmul64(nu[j], (int64_t)pn2), wmul64(pnu[j], cdp));
                  for (int j = 0; j < 3; ++j) cxv[j] = wsub64(cp[j], wadd64(np[j], wdiv64(wmul64(pn[j], cdp), (int64_t)pn2)));
                  const uint64_t cx2 = (uint64_t)wadd64(wadd64(wmul64(cxv[0], cxv[0]), wmul64(cxv[1], cxv[1])), wmul64(cxv[2], cxv[2]));
                  const uint64_t norm = int_sqrt(cx2 * pn2);
                  const int64_t cx_uv[2] = {wmul64(pnu[1], (int64_t)norm), wmul64((int64_t)(0 - (uint64_t)pnu[0]), (int64_t)norm)};
                  if (next_bit >= bits.size()) return host_fail(DMI_ERR_ENTROPY, "orientation stream exhausted");
                  const bool first = bits[next_bit++] != 0;                       // the encoder's choice, read back
                  for (int j = 0; j < 2; ++j) pred[j] = (int32_t)wdiv64(first ? wadd64(x_uv[j], cx_uv[j]) : wsub64(x_uv[j], cx_uv[j]), (int64_t)pn2);
                  done = true;
                }
              }
            }
          }
          if (!done) { if (have(nv, k)) { pred[0] = val[(size_t)nv * 2]; pred[1] = val[(size_t)nv * 2 + 1]; } else previous_value(); }
        } else {
          return host_fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "unknown prediction scheme " + std::to_string(a.scheme));
        }
        const uint32_t v = t.corner_to_vertex[c];
        for (int j = 0; j < N; ++j) {
          int32_t o;
          if (a.transform == 1) {                                                 // wrapped_difference.rs:54-99 inverted
            int32_t p = pred[j];
            if (t_min <= t_max) p = p < t_min ? t_min : (p > t_max ? t_max : p);
            o = wadd(p, from_positive(sym[k * N + j]));
            if (o > t_max) o = wsub(o, max_diff); else if (o < t_min) o = wadd(o, max_diff);
          } else if (a.transform == 0) {                                          // difference.rs:26-34 inverted
            o = wadd(pred[j], from_positive(sym[k * N + j]));
          } else {
            return host_fail(DMI_ERR_UNSUPPORTED_DATA_TYPE, "transform " + std::to_string(a.transform) + " on a sequential scheme");
          }
          val[(size_t)v * N + j] = o;
        }
        when[v] = (uint32_t)k + 1;
      }
      if (a.scheme == 5 && next_bit != bits.size()) return host_fail(DMI_ERR_ENTROPY, "orientation bits left over");
      inverse_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_att0).count();
      if ((rc = d_val.upload(val.data(), val.size() * 4, s))) return rc;
      if (a.att_type == DMI_ATT_POSITION && N == 3 && pos_att < 0) {   // the parent of the normals / texture coordinates that follow
        pos_att = (int)i;
        pos_by_vertex = val;
        if ((rc = d_pos.upload(pos_by_vertex.data(), pos_by_vertex.size() * 4, s)) || (rc = d_c2v_pos.upload(t.corner_to_vertex, C * 4, s))) return rc;
      }
    }
    // ---- dequantization, scattered to the points (device) ----
    DevBuf d_out;
    owner->values[i].assign((size_t)num_points * a.num_components, 0.0f);
    if ((rc = d_out.alloc(owner->values[i].size() * 4))) return rc;
    HIP_TRY_D(hipMemsetAsync(d_out.p, 0, owner->values[i].size() * 4, s));
    DequantizeArgs qa{};
    qa.c2p = d_c2p.as<uint32_t>(); qa.c2v = d_c2v.as<uint32_t>(); qa.corners = C; qa.q = d_val.as<int32_t>(); qa.out = d_out.as<float>(); qa.last_corner = d_last.as<uint32_t>();
    for (int k = 0; k < 4; ++k) qa.mn[k] = q_min[k];
    qa.delta = port[i] == 2 ? q_range / (float)(uint64_t)((1ull << q_bits) - 1ull) : 0.0f;
    qa.kind = port[i]; qa.N = a.num_components;
    launch_dequantize(qa, s);
    HIP_TRY_D(hipMemcpyAsync(owner->values[i].data(), d_out.p, owner->values[i].size() * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY_D(hipStreamSynchronize(s));   // (the attribute's device arrays go out of scope)
    a.values = owner->values[i].data();
  }
  {
    const auto t_end = std::chrono::steady_clock::now();
    g_last_decode.inverse_ms = (float)inverse_ms;
    g_last_decode.device_ms = std::chrono::duration<float, std::milli>(t_end - t_inv0).count() - (float)inverse_ms;
    g_last_decode.attributes_ms = std::chrono::duration<float, std::milli>(t_end - t_call0).count();
  }
  out->num_attributes = n_atts;
  out->attributes = owner->atts.data();
  out->owner = owner.release();
  return DMI_OK;
}

}  // extern "C"
