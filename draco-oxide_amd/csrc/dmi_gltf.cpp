// dmi_gltf.cpp — dmi_transcode_assets: a LIST of glTF assets in, their KHR_draco_mesh_compression GLBs out, with nothing but this library in
// between (round 5).  What io/gltf/transcoder.rs:134-151 does per file — read_scene (io/gltf/decode.rs:2328-2525: one Mesh per triangle
// primitive, attributes in sorted semantic order, accessors read as raw little-endian f32 with the view's stride, NORMAL / TEXCOORD_0 as Corner
// attributes whose parent is the position), compress_scene (io/gltf/encode.rs:932-955: encode::encode per primitive) and write_scene
// (io/gltf/encode.rs:958-1097,362-400: blob appended to the BIN chunk and zero-padded to 4 bytes with the pad inside the bufferView, placeholder
// accessors, the extension's attribute ids, GLB container with a space-padded JSON chunk) — for ALL files of the list at once:
//   caller's thread   container + JSON parse, primitives planned, accessor descriptors made (bounds-checked views of the caller's bytes — copied up
//                     where they lie when the caller read its files into dmi_host_alloc memory, packed by host threads otherwise: dmi_hostmem.cpp),
//                     pushed into a dmi_transcoder per device (the least loaded one takes the next primitive)
//   library threads   per device: build ∥ prepare ∥ encode of consecutive stages (dmi_transcode.cpp)
//   assembly threads  a file is written — JSON patched and serialised, BIN chunk laid out, blobs copied in — as soon as its last primitive is final,
//                     into a recycled output arena
// Until round 4 this loop was Python (draco-oxide_amd/gltf.py: parse 10 ms + views 9 ms + reassembly 35–45 ms per 1024 files under one interpreter
// lock, a third of the call); gltf.py keeps the same steps as the reference for the tests and drives the rank-sharded form.
// The embedded .drc blobs are the bit-exact contract; the JSON is the input's document with the same edits gltf.py makes.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/draco_mi.h"
#include "dmi_host.hpp"
#include "dmi_json.hpp"

using namespace dmi;
using json::Value;

namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- output arena: blocks recycled across calls (81 MB of fresh pages per 1024 files cost more than writing them) ----
struct Block { uint8_t* p = nullptr; size_t cap = 0; };
struct BlockPool {
  std::mutex m;
  std::vector<Block> free_;
  size_t held = 0;
  Block take(size_t n) {
    {
      std::lock_guard<std::mutex> lock(m);
      size_t best = free_.size();
      for (size_t k = 0; k < free_.size(); ++k) if (free_[k].cap >= n && (best == free_.size() || free_[k].cap < free_[best].cap)) best = k;
      if (best != free_.size()) { Block b = free_[best]; free_.erase(free_.begin() + (long)best); held -= b.cap; return b; }
    }
    Block b;
    const size_t cap = (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* q = nullptr;
    if (posix_memalign(&q, (size_t)2 << 20, cap) != 0 || !q) return b;
    advise_huge_pages(q, cap);
    b.p = static_cast<uint8_t*>(q); b.cap = cap;
    return b;
  }
  void give(Block b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> lock(m);
    if (held + b.cap <= host_pool_limit() / 4) { held += b.cap; free_.push_back(b); } else std::free(b.p);
  }
  void drop_all() { std::lock_guard<std::mutex> lock(m); for (auto& b : free_) std::free(b.p); free_.clear(); held = 0; }
};
BlockPool& block_pool() { static BlockPool p; return p; }
constexpr size_t kBlockBytes = (size_t)32 << 20;

struct Arena {
  std::mutex m;
  std::vector<Block> blocks;
  size_t used = 0;   // of the last block
  uint8_t* reserve(size_t n) {
    std::lock_guard<std::mutex> lock(m);
    if (blocks.empty() || used + n > blocks.back().cap) {
      Block b = block_pool().take(std::max(n, kBlockBytes));
      if (!b.p) return nullptr;
      blocks.push_back(b);
      used = 0;
    }
    uint8_t* p = blocks.back().p + used;
    used += (n + 63) & ~(size_t)63;
    return p;
  }
  ~Arena() { for (auto& b : blocks) block_pool().give(b); }
};

struct Span { const uint8_t* p = nullptr; size_t n = 0; };

struct Prim {
  Value* node = nullptr;                 // the primitive's JSON object
  std::vector<std::string> names;        // attribute names in AttributeId order
  uint64_t triangles = 0;
  int device_slot = -1;                  // which transcoder took it
  uint32_t push_index = 0;               // its index there
};

struct Asset {
  Value doc;
  std::vector<Span> buffers;
  std::vector<Prim> prims;               // the primitives that get compressed
  std::vector<dmi_raw_mesh> raws;        // … as the transcoder takes them (views of the caller's bytes and of `owned`)
  std::deque<std::vector<dmi_raw_accessor>> accessors;   // stable storage for the descriptors pushed
  std::deque<std::vector<uint32_t>> owned;               // converted feature ids / generated indices / parent ids
  std::atomic<uint32_t> left{0};         // primitives not final yet
  // result
  const uint8_t* out = nullptr;
  size_t out_bytes = 0;
  std::vector<std::pair<size_t, size_t>> blobs;   // (offset in the file, bytes) per compressed primitive, in primitive order
};

struct PerDevice {
  dmi_transcoder* t = nullptr;
  uint64_t load = 0;                                   // triangles dealt to it
  uint64_t triangles = 0; uint32_t primitives = 0;     // (its pusher's counts)
  std::mutex owner_mutex;                              // (the transcoder's encode thread reads `owner` while the caller's thread appends)
  std::vector<std::pair<uint32_t, uint32_t>> owner;    // push index → (asset, primitive)
  struct dmi_transcoded* self = nullptr;
  int slot = 0;
};

int gfail(const std::string& what) { return host_fail(DMI_ERR_INVALID_ARGUMENT, "gltf: " + what); }

const Value* member(const Value& v, const char* key) { return v.find(key); }
bool index_of(const Value* v, uint64_t* out) { return v && v->as_index(out); }

const char* const kStandardPrefixes[] = {"POSITION", "NORMAL", "TANGENT", "TEXCOORD_", "COLOR_", "JOINTS_", "WEIGHTS_"};
bool starts_with(const std::string& s, const char* p) { return s.compare(0, std::strlen(p), p) == 0; }
bool is_standard(const std::string& k) { for (const char* p : kStandardPrefixes) if (starts_with(k, p)) return true; return false; }
int semantic_type(const std::string& k) { return k == "POSITION" ? DMI_ATT_POSITION : k == "NORMAL" ? DMI_ATT_NORMAL : k == "TEXCOORD_0" ? DMI_ATT_TEXCOORD : -1; }
int components_of(const std::string& t) { return t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : 0; }

// GLB container (io/gltf/encode.rs:362-400 writes it; the importer reads the same): "glTF", 2, length | chunks (length, type, payload)
int read_glb(const uint8_t* data, size_t n, Span* js, Span* bin) {
  if (n < 12 || std::memcmp(data, "glTF", 4) != 0) return gfail("not a GLB file");
  uint32_t version, length;
  std::memcpy(&version, data + 4, 4); std::memcpy(&length, data + 8, 4);
  if (version != 2) return gfail("not a GLB v2 file");
  if (length > n) return gfail("GLB length field exceeds the buffer");
  size_t off = 12;
  while (off + 8 <= length) {
    uint32_t clen, ctype;
    std::memcpy(&clen, data + off, 4); std::memcpy(&ctype, data + off + 4, 4);
    if ((size_t)clen > length - off - 8) return gfail("GLB chunk exceeds the file");
    if (ctype == 0x4E4F534Au) *js = Span{data + off + 8, clen};
    else if (ctype == 0x004E4942u) *bin = Span{data + off + 8, clen};
    off += 8 + (size_t)clen;
  }
  if (!js->p) return gfail("GLB without a JSON chunk");
  return DMI_OK;
}

struct AccessorView { const uint8_t* p = nullptr; uint64_t count = 0; uint32_t stride = 0; uint32_t component_type = 0; int components = 0; };

// accessor `index` of the document as bytes of its buffer: elem_bytes per element (0: 4 · components — raw f32 rows), the view's byteStride or tight.
// Every offset is checked against the bufferView's own byteLength AND the buffer (sums in 64 bits with overflow tests: a crafted offset must not wrap back
// into range); a stride below the element size or off the 4-byte grid is refused.
int view_accessor(const Asset& a, uint64_t index, size_t elem_bytes, bool use_stride, AccessorView* out) {
  const Value* accs = member(a.doc, "accessors");
  if (!accs || !accs->is_array() || index >= accs->items.size()) return gfail("accessor index out of range");
  const Value& acc = accs->items[index];
  const std::string who = "accessor " + std::to_string(index);
  uint64_t bv, count, off_a = 0, off_v = 0, stride = 0, buf = 0, ct = 0, view_len = 0;
  if (!index_of(member(acc, "bufferView"), &bv)) return gfail(who + " without a bufferView");
  if (!index_of(member(acc, "count"), &count)) return gfail(who + " without a count");
  if (member(acc, "byteOffset") && !index_of(member(acc, "byteOffset"), &off_a)) return gfail("accessor byteOffset");
  (void)index_of(member(acc, "componentType"), &ct);
  const Value* ty = member(acc, "type");
  const int comps = ty && ty->kind == Value::String ? components_of(ty->text) : 0;
  const Value* views = member(a.doc, "bufferViews");
  if (!views || !views->is_array() || bv >= views->items.size()) return gfail("bufferView index out of range");
  const Value& view = views->items[bv];
  if (member(view, "byteOffset") && !index_of(member(view, "byteOffset"), &off_v)) return gfail("bufferView byteOffset");
  if (member(view, "byteStride") && !index_of(member(view, "byteStride"), &stride)) return gfail("bufferView byteStride");
  if (member(view, "buffer") && !index_of(member(view, "buffer"), &buf)) return gfail("bufferView buffer");
  if (!index_of(member(view, "byteLength"), &view_len)) return gfail("bufferView " + std::to_string(bv) + " without a byteLength");
  if (buf >= a.buffers.size()) return gfail("buffer index out of range");
  if (!elem_bytes) { if (!comps) return gfail(who + ": unknown type"); elem_bytes = 4u * (size_t)comps; }
  if (!use_stride || !stride) stride = elem_bytes;
  if (stride < elem_bytes) return gfail(who + ": byteStride below the element size");
  if (stride > (1u << 20) || count >= (1ull << 32)) return gfail(who + ": stride / count out of range");
  const Span& b = a.buffers[buf];
  if (off_v > b.n || view_len > b.n - off_v) return gfail("bufferView " + std::to_string(bv) + " reaches past its buffer");
  // extent of the accessor inside its view: off_a + stride·(count − 1) + elem_bytes ≤ view_len (no sum can wrap: count < 2^32, stride ≤ 2^20)
  const uint64_t extent = count ? stride * (count - 1) + elem_bytes : 0;
  if (off_a > view_len || extent > view_len - off_a) return gfail(who + " reaches past its bufferView");
  out->p = b.p + off_v + off_a; out->count = count; out->stride = (uint32_t)stride; out->component_type = (uint32_t)ct; out->components = comps;
  return DMI_OK;
}

// Rust's `f32 as u32`: truncates, saturates, NaN → 0 (io/gltf/decode.rs:2528-2584 reads a FLOAT feature-id accessor that way)
uint32_t f32_as_u32(float f) { return !(f == f) || f <= 0.0f ? 0u : f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f; }

}  // namespace

struct dmi_transcoded {
  std::vector<std::unique_ptr<Asset>> assets;   // (reserved up front: the assembly threads index it while the caller's thread appends)
  std::vector<std::unique_ptr<PerDevice>> devs;
  Arena arena;
  dmi_transcode_stats stats{};
  // assembly
  std::mutex q_mutex;
  std::condition_variable q_cv;
  std::deque<uint32_t> ready;          // assets whose primitives are all final
  bool q_closed = false;
  std::mutex err_mutex;
  int rc = DMI_OK;
  std::string err;
  std::atomic<uint64_t> assemble_ns{0};
  void fail_with(int code, const std::string& what) { std::lock_guard<std::mutex> lock(err_mutex); if (rc == DMI_OK) { rc = code; err = what; } }
  void enqueue(uint32_t asset) { { std::lock_guard<std::mutex> lock(q_mutex); ready.push_back(asset); } q_cv.notify_one(); }
  ~dmi_transcoded() {
    for (auto& d : devs) if (d->t) dmi_transcoder_destroy(d->t);
  }
};

namespace {

// ---- planning: the primitives of a document that get compressed (gltf.py _plan / primitive_to_raw; decode.rs:2328-2525) ----
int plan_asset(Asset& a) {
  Value* meshes = a.doc.find("meshes");
  if (!meshes || !meshes->is_array()) return DMI_OK;
  for (Value& mesh : meshes->items) {
    Value* prims = mesh.find("primitives");
    if (!prims || !prims->is_array()) continue;
    for (Value& prim : prims->items) {
      uint64_t mode = 4;
      if (const Value* m = member(prim, "mode")) if (!m->as_index(&mode)) mode = ~0ull;
      const Value* atts = member(prim, "attributes");
      if (mode != 4 || !atts || !atts->is_object() || !atts->find("POSITION")) continue;
      if (const Value* ext = member(prim, "extensions")) if (ext->find("KHR_draco_mesh_compression")) return gfail("KHR_draco_mesh_compression input is not supported (decode.rs:2478-2483)");
      // the reference sorts EVERY standard semantic by name (decode.rs:2410) and hands the position's index in THAT list to NORMAL / TEXCOORD_0 as
      // their parent id (:2414-2425), but adds only POSITION, NORMAL and TEXCOORD_0 (:2431-2473): with a COLOR_n / JOINTS_n in front of POSITION
      // the parent id names the wrong attribute and its encoder panics — refused here like in gltf.py
      std::vector<std::string> standard;
      for (const auto& m : atts->members) if (is_standard(m.first)) standard.push_back(m.first);
      std::sort(standard.begin(), standard.end());
      if (std::adjacent_find(standard.begin(), standard.end()) != standard.end()) return gfail("a primitive names the attribute " + *std::adjacent_find(standard.begin(), standard.end()) + " twice");
      Prim p;
      p.node = &prim;
      for (const auto& k : standard) if (semantic_type(k) >= 0) p.names.push_back(k);
      const size_t pos_id = (size_t)(std::find(standard.begin(), standard.end(), "POSITION") - standard.begin());
      const size_t pos_at = (size_t)(std::find(p.names.begin(), p.names.end(), "POSITION") - p.names.begin());
      if (pos_id != pos_at && p.names.size() > 1) return gfail("the reference hands NORMAL / TEXCOORD_0 a parent id that is not the position attribute for this set of semantics; its encoder panics on such a primitive");
      std::vector<std::string> feat;
      for (const auto& m : atts->members) if (starts_with(m.first, "_FEATURE_ID_")) feat.push_back(m.first);
      std::sort(feat.begin(), feat.end());   // (the reference walks a HashMap here: its order changes from run to run; name order like gltf.py)
      if (std::adjacent_find(feat.begin(), feat.end()) != feat.end()) return gfail("a primitive names the attribute " + *std::adjacent_find(feat.begin(), feat.end()) + " twice");
      p.names.insert(p.names.end(), feat.begin(), feat.end());
      a.prims.push_back(std::move(p));
    }
  }
  return DMI_OK;
}

// one planned primitive as a dmi_raw_mesh over the caller's bytes
int raw_of(Asset& a, Prim& p, dmi_raw_mesh* out) {
  const Value* atts = member(*p.node, "attributes");
  a.accessors.emplace_back();
  std::vector<dmi_raw_accessor>& acc = a.accessors.back();
  uint32_t pos_id = 0;
  for (uint32_t i = 0; i < p.names.size(); ++i) if (p.names[i] == "POSITION") pos_id = i;
  a.owned.emplace_back(1, pos_id);
  const uint32_t* parent = a.owned.back().data();
  uint64_t count = 0;
  for (const std::string& name : p.names) {
    uint64_t ai;
    if (!index_of(atts->find(name.c_str()), &ai)) return gfail("attribute " + name + ": not an accessor index");
    dmi_raw_accessor r{};
    const int ty = semantic_type(name);
    if (ty >= 0) {
      AccessorView v;
      if (int rc = view_accessor(a, ai, 0, true, &v)) return rc;
      // the reference reads these three as raw f32 rows whatever the accessor says (decode.rs:2277-2309); KHR_mesh_quantization / normalized-integer
      // inputs would come out as valid-looking garbage blobs: refused
      if (v.component_type != 5126) return gfail("attribute " + name + ": componentType " + std::to_string(v.component_type) + " (only FLOAT accessors are transcoded)");
      if (v.components != (ty == DMI_ATT_TEXCOORD ? 2 : 3)) return gfail("attribute " + name + ": accessor type is not " + (ty == DMI_ATT_TEXCOORD ? "VEC2" : "VEC3"));
      if (v.stride & 3u) return gfail("attribute " + name + ": byteStride is not a multiple of 4");
      count = v.count;
      r.data = v.p; r.count = (uint32_t)v.count; r.byte_stride = v.stride == 4u * (uint32_t)v.components ? 0u : v.stride;
      r.component_type = DMI_F32; r.num_components = (uint8_t)v.components; r.att_type = (uint8_t)ty;
      r.domain = ty == DMI_ATT_POSITION ? DMI_DOMAIN_POSITION : DMI_DOMAIN_CORNER;
      if (ty != DMI_ATT_POSITION) { r.num_parents = 1; r.parents = parent; }
    } else {   // _FEATURE_ID_n → Custom u32 corner attribute without parents (decode.rs:2490-2516)
      const Value* accs = member(a.doc, "accessors");
      uint64_t ct = 0;
      if (!accs || !accs->is_array() || ai >= accs->items.size()) return gfail("accessor index out of range");
      (void)index_of(member(accs->items[ai], "componentType"), &ct);
      const size_t eb = ct == 5121 ? 1 : ct == 5123 ? 2 : ct == 5125 || ct == 5126 ? 4 : 0;
      if (!eb) return gfail("unsupported component type " + std::to_string(ct) + " for a feature-id attribute");
      AccessorView v;
      if (int rc = view_accessor(a, ai, eb, true, &v)) return rc;
      a.owned.emplace_back((size_t)v.count);
      std::vector<uint32_t>& ids = a.owned.back();
      for (uint64_t k = 0; k < v.count; ++k) {
        const uint8_t* e = v.p + k * v.stride;
        if (eb == 1) ids[k] = *e;
        else if (eb == 2) { uint16_t x; std::memcpy(&x, e, 2); ids[k] = x; }
        else if (ct == 5125) { uint32_t x; std::memcpy(&x, e, 4); ids[k] = x; }
        else { float f; std::memcpy(&f, e, 4); ids[k] = f32_as_u32(f); }
      }
      r.data = ids.data(); r.count = (uint32_t)v.count; r.byte_stride = 0;
      r.component_type = DMI_U32; r.num_components = 1; r.att_type = DMI_ATT_CUSTOM; r.domain = DMI_DOMAIN_CORNER;
    }
    acc.push_back(r);
  }
  dmi_raw_mesh m{};
  m.atts = acc.data(); m.n_atts = (uint32_t)acc.size();
  if (const Value* ind = member(*p.node, "indices")) {
    uint64_t ii;
    if (!ind->as_index(&ii)) return gfail("indices: not an accessor index");
    const Value* accs = member(a.doc, "accessors");
    uint64_t ct = 0;
    if (!accs || !accs->is_array() || ii >= accs->items.size()) return gfail("accessor index out of range");
    (void)index_of(member(accs->items[ii], "componentType"), &ct);
    const size_t eb = ct == 5121 ? 1 : ct == 5123 ? 2 : ct == 5125 ? 4 : 0;
    if (!eb) return gfail("index accessor: component type " + std::to_string(ct));
    AccessorView v;
    if (int rc = view_accessor(a, ii, eb, false, &v)) return rc;
    m.indices = v.p; m.index_type = eb == 1 ? DMI_U8 : eb == 2 ? DMI_U16 : DMI_U32; m.num_faces = (uint32_t)(v.count / 3);
  } else {   // no index accessor: the points in order
    a.owned.emplace_back((size_t)count);
    std::vector<uint32_t>& seq = a.owned.back();
    for (uint64_t k = 0; k < count; ++k) seq[k] = (uint32_t)k;
    m.indices = seq.data(); m.index_type = DMI_U32; m.num_faces = (uint32_t)(count / 3);
  }
  p.triangles = m.num_faces;
  *out = m;
  return DMI_OK;
}

// ---- assembly of one file (gltf.py _assemble + write_glb; encode.rs:958-1097,362-400) ----
struct Piece { const uint8_t* p; size_t n; const uint8_t* p2; size_t n2; };   // (a blob is two pieces back to back: header + connectivity, attribute section)

// How many places of the document name each accessor: attributes, indices and morph targets of EVERY primitive, animation samplers, skins (gltf.py _accessor_users)
void count_accessor_users(const Value& doc, std::vector<uint32_t>& users) {
  auto use = [&](const Value* v) { uint64_t i; if (v && v->as_index(&i) && i < users.size()) ++users[i]; };
  if (const Value* meshes = member(doc, "meshes")) if (meshes->is_array())
    for (const Value& mesh : meshes->items) {
      const Value* prims = member(mesh, "primitives");
      if (!prims || !prims->is_array()) continue;
      for (const Value& prim : prims->items) {
        if (const Value* atts = member(prim, "attributes")) if (atts->is_object()) for (const auto& m : atts->members) use(&m.second);
        use(member(prim, "indices"));
        if (const Value* tg = member(prim, "targets")) if (tg->is_array()) for (const Value& t : tg->items) if (t.is_object()) for (const auto& m : t.members) use(&m.second);
      }
    }
  if (const Value* anims = member(doc, "animations")) if (anims->is_array())
    for (const Value& an : anims->items) if (const Value* smp = member(an, "samplers")) if (smp->is_array())
      for (const Value& sp : smp->items) { use(member(sp, "input")); use(member(sp, "output")); }
  if (const Value* skins = member(doc, "skins")) if (skins->is_array()) for (const Value& sk : skins->items) use(member(sk, "inverseBindMatrices"));
}

// A compressed primitive's accessors become placeholders (no bufferView, new counts).  An accessor that something ELSE names too — a primitive that stays
// uncompressed (another mode, no face left), a second compressed primitive, a morph target, an animation — must keep its data and must not take another
// primitive's counts: the compressed primitive gets a copy of its own at the end of the accessor list (the reference writes fresh accessors per primitive:
// encode.rs:958-1097).  compressed[k]: primitive k of a.prims has a face left.
void privatize_accessors(Asset& a, const std::vector<uint8_t>& compressed) {
  Value* accs = a.doc.find("accessors");
  if (!accs || !accs->is_array()) return;
  std::vector<uint32_t> users(accs->items.size(), 0);
  count_accessor_users(a.doc, users);
  for (size_t k = 0; k < a.prims.size(); ++k) {
    if (!compressed[k]) continue;
    Prim& p = a.prims[k];
    Value* atts = p.node->find("attributes");
    auto own = [&](Value* ref) {
      uint64_t ai;
      if (!ref || !ref->as_index(&ai) || ai >= users.size() || users[ai] <= 1) return;
      --users[ai];
      Value copy = accs->items[ai];
      accs->items.push_back(std::move(copy));
      *ref = Value::number(accs->items.size() - 1);
    };
    if (atts && atts->is_object()) for (const std::string& n : p.names) own(atts->find(n.c_str()));
    own(p.node->find("indices"));
  }
}

int assemble(dmi_transcoded& R, Asset& a) {
  struct Res { dmi_buffer head{}, section{}; uint32_t nf = 0, np = 0; };
  std::vector<Res> res(a.prims.size());
  std::vector<uint8_t> compressed(a.prims.size(), 0);
  for (size_t k = 0; k < a.prims.size(); ++k) {
    Prim& p = a.prims[k];
    PerDevice& d = *R.devs[(size_t)p.device_slot];
    if (int rc = dmi_transcoder_result(d.t, p.push_index, &res[k].head, &res[k].section, &res[k].nf, &res[k].np)) return rc;
    compressed[k] = res[k].nf ? 1 : 0;
  }
  privatize_accessors(a, compressed);
  Value* accs = a.doc.find("accessors");
  const size_t n_acc = accs && accs->is_array() ? accs->items.size() : 0;
  std::vector<uint8_t> replaced(n_acc, 0);
  for (size_t k = 0; k < a.prims.size(); ++k) {
    Prim& p = a.prims[k];
    if (!res[k].nf) continue;   // no face left: the reference leaves such a primitive alone (encode.rs:934-936)
    const Value* atts = member(*p.node, "attributes");
    for (const std::string& n : p.names) { uint64_t ai; if (index_of(atts->find(n.c_str()), &ai) && ai < n_acc) replaced[ai] = 1; }
    uint64_t ii;
    if (index_of(member(*p.node, "indices"), &ii) && ii < n_acc) replaced[ii] = 1;
  }
  std::vector<Piece> pieces;
  size_t size = 0;
  Value new_views = Value::array();
  std::vector<int64_t> view_map;
  const Value* old_views = member(a.doc, "bufferViews");
  const size_t n_views = old_views && old_views->is_array() ? old_views->items.size() : 0;
  view_map.assign(n_views, -1);
  auto put = [&](Piece pc) { pieces.push_back(pc); size += pc.n + pc.n2; size = (size + 3) & ~(size_t)3; };
  int rc = DMI_OK;
  auto carry = [&](uint64_t vi, uint64_t* out) -> bool {
    if (vi >= n_views) { rc = gfail("bufferView index out of range"); return false; }
    if (view_map[vi] < 0) {
      Value v = old_views->items[vi];
      uint64_t start = 0, len = 0, buf = 0;
      if ((member(v, "byteOffset") && !index_of(member(v, "byteOffset"), &start)) || !index_of(member(v, "byteLength"), &len) || (member(v, "buffer") && !index_of(member(v, "buffer"), &buf)) ||
          buf >= a.buffers.size() || start > a.buffers[buf].n || len > a.buffers[buf].n - start) { rc = gfail("bufferView " + std::to_string(vi) + " reaches past its buffer"); return false; }
      v.set("byteOffset", Value::number(size));
      v.set("buffer", Value::number(0));
      put(Piece{a.buffers[buf].p + start, (size_t)len, nullptr, 0});
      view_map[vi] = (int64_t)new_views.items.size();
      new_views.items.push_back(std::move(v));
    }
    *out = (uint64_t)view_map[vi];
    return true;
  };
  for (size_t i = 0; i < n_acc; ++i) {
    Value& acc = accs->items[i];
    if (replaced[i]) { acc.erase("bufferView"); acc.erase("byteOffset"); }
    else if (Value* bv = acc.find("bufferView")) { uint64_t vi, nv; if (!bv->as_index(&vi)) return gfail("accessor bufferView"); if (!carry(vi, &nv)) return rc; *bv = Value::number(nv); }
  }
  if (Value* images = a.doc.find("images")) if (images->is_array())
    for (Value& img : images->items) if (Value* bv = img.find("bufferView")) { uint64_t vi, nv; if (!bv->as_index(&vi)) return gfail("image bufferView"); if (!carry(vi, &nv)) return rc; *bv = Value::number(nv); }
  bool any = false;
  std::vector<std::pair<size_t, size_t>> spans;   // in the BIN chunk
  for (size_t k = 0; k < a.prims.size(); ++k) {
    if (!res[k].nf) continue;
    Prim& p = a.prims[k];
    any = true;
    const size_t start = size;
    put(Piece{res[k].head.data, res[k].head.len, res[k].section.data, res[k].section.len});
    spans.emplace_back(start, res[k].head.len + res[k].section.len);
    Value bv = Value::object();
    bv.set("buffer", Value::number(0)); bv.set("byteOffset", Value::number(start)); bv.set("byteLength", Value::number(size - start));   // (the length includes the pad: encode.rs:958-967)
    new_views.items.push_back(std::move(bv));
    // AttributeId = add order = `names` order (the built mesh has Position in slot 0, ids unchanged: builder.rs:115-125)
    Value ext = Value::object(), ids = Value::object();
    for (size_t q = 0; q < p.names.size(); ++q) ids.set(p.names[q], Value::number(q));
    ext.set("bufferView", Value::number(new_views.items.size() - 1));
    ext.set("attributes", std::move(ids));
    Value* exts = p.node->find("extensions");
    if (!exts || !exts->is_object()) exts = &p.node->set("extensions", Value::object());
    exts->set("KHR_draco_mesh_compression", std::move(ext));
    uint64_t ii;
    if (index_of(member(*p.node, "indices"), &ii) && ii < n_acc) accs->items[ii].set("count", Value::number((uint64_t)res[k].nf * 3));
    const Value* atts = member(*p.node, "attributes");
    for (const std::string& n : p.names) { uint64_t ai; if (index_of(atts->find(n.c_str()), &ai) && ai < n_acc) accs->items[ai].set("count", Value::number(res[k].np)); }
  }
  a.doc.set("bufferViews", std::move(new_views));
  {
    Value bufs = Value::array(), b0 = Value::object();
    b0.set("byteLength", Value::number(size));
    bufs.items.push_back(std::move(b0));
    a.doc.set("buffers", std::move(bufs));
  }
  if (any) for (const char* key : {"extensionsUsed", "extensionsRequired"}) {
    Value* lst = a.doc.find(key);
    if (!lst || !lst->is_array()) lst = &a.doc.set(key, Value::array());
    bool have = false;
    for (const Value& s : lst->items) have = have || (s.kind == Value::String && s.text == "KHR_draco_mesh_compression");
    if (!have) lst->items.push_back(Value::string("KHR_draco_mesh_compression"));
  }
  std::string js;
  js.reserve(4096);
  json::write(a.doc, js);
  while (js.size() % 4) js.push_back(' ');   // the JSON chunk is space padded (encode.rs:392-396)
  const size_t n_bin = size;
  const size_t total = 12 + 8 + js.size() + (n_bin ? 8 + n_bin : 0);
  if (total >= (1ull << 32)) return gfail("output GLB of 4 GiB or more");
  uint8_t* o = R.arena.reserve(total);
  if (!o) return host_fail(DMI_ERR_OUT_OF_MEMORY, "output arena");
  uint8_t* w = o;
  auto u32 = [&](uint32_t v) { std::memcpy(w, &v, 4); w += 4; };
  std::memcpy(w, "glTF", 4); w += 4; u32(2); u32((uint32_t)total);
  u32((uint32_t)js.size()); u32(0x4E4F534Au);
  std::memcpy(w, js.data(), js.size()); w += js.size();
  if (n_bin) {
    u32((uint32_t)n_bin); u32(0x004E4942u);
    uint8_t* bin0 = w;
    for (const Piece& pc : pieces) {
      if (pc.n) std::memcpy(w, pc.p, pc.n);
      w += pc.n;
      if (pc.n2) std::memcpy(w, pc.p2, pc.n2);
      w += pc.n2;
      while ((size_t)(w - bin0) % 4) *w++ = 0;
    }
    for (auto& s : spans) a.blobs.emplace_back((size_t)(bin0 - o) + s.first, s.second);
  }
  a.out = o; a.out_bytes = total;
  // the document, the descriptors and the converted arrays have done their work: released here, on this (assembly) thread, not when the caller drops the result
  a.doc = Value();
  a.prims.clear(); a.prims.shrink_to_fit();
  a.raws.clear(); a.raws.shrink_to_fit();
  a.accessors.clear(); a.owned.clear();
  return DMI_OK;
}

void assemble_loop(dmi_transcoded* R) {
  for (;;) {
    uint32_t ai;
    {
      std::unique_lock<std::mutex> lock(R->q_mutex);
      R->q_cv.wait(lock, [&] { return !R->ready.empty() || R->q_closed; });
      if (R->ready.empty()) return;
      ai = R->ready.front();
      R->ready.pop_front();
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = assemble(*R, *R->assets[ai])) R->fail_with(rc, dmi_last_error());
    R->assemble_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
}

// a transcoder reports the primitives [first, first + count) of its push order final (called from its encode thread)
void on_done(void* user, uint32_t first, uint32_t count) {
  PerDevice* d = static_cast<PerDevice*>(user);
  dmi_transcoded* R = d->self;
  for (uint32_t k = first; k < first + count; ++k) {
    uint32_t ai;
    { std::lock_guard<std::mutex> lock(d->owner_mutex); ai = d->owner[k].first; }
    if (R->assets[ai]->left.fetch_sub(1) == 1) R->enqueue(ai);
  }
}

}  // namespace

extern "C" {

int dmi_transcode_assets(const dmi_gltf_asset* assets, uint32_t n, const dmi_config* cfg, const int32_t* devices, uint32_t n_devices, uint32_t flags, dmi_transcoded** out) {
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
  (void)flags;
  if (!out || (!assets && n)) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  *out = nullptr;
  const double t_start = now_ms();
  std::unique_ptr<dmi_transcoded> R(new dmi_transcoded());
  const int ndev_visible = dmi_device_count();
  if (ndev_visible <= 0) return host_fail(DMI_ERR_NO_DEVICE, "no HIP device visible; libdraco_mi has no CPU fallback");
  std::vector<int32_t> devs;
  if (devices && n_devices) devs.assign(devices, devices + n_devices); else devs.push_back(cfg ? cfg->device : 0);
  for (int32_t d : devs) if (d < 0 || d >= ndev_visible) return host_fail(DMI_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  const size_t ND = devs.size();
  // ---- the files are dealt to the devices BEFORE anything is parsed (round 6), by their size in bytes — what stands for a file's triangle count until its
  //      JSON is read: largest first, each to the device with the least bytes so far (LPT).  A device's transcoder then starts on its first files while the
  //      others are still being parsed (by a small pool of threads: files are independent), and the pipeline's LAST stage — whose prepare and encode nothing
  //      overlaps — is made of the smallest meshes.  Results are placed by file index: the order of the output does not depend on any of this.
  auto bytes_of = [&](uint32_t i) { size_t b = assets[i].glb ? assets[i].glb_bytes : assets[i].json_bytes; for (uint32_t k = 0; !assets[i].glb && k < assets[i].n_buffers; ++k) b += assets[i].buffers[k].bytes; return b; };
  uint64_t in_bytes = 0;
  std::vector<uint32_t> order(n);
  for (uint32_t i = 0; i < n; ++i) { order[i] = i; in_bytes += bytes_of(i); }
  if (!dbg_on(DMI_DBG_FILE_ORDER)) std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return bytes_of(x) > bytes_of(y); });
  std::vector<std::vector<uint32_t>> dev_files(ND);
  std::vector<uint64_t> dev_bytes(ND, 0);
  for (uint32_t i : order) {
    size_t best = 0;
    for (size_t s = 1; s < ND; ++s) if (dev_bytes[s] < dev_bytes[best]) best = s;
    dev_files[best].push_back(i);
    dev_bytes[best] += bytes_of(i) + 1;
  }
  if (dbg_on(DMI_DBG_SMALL_HEAD)) {   // (experiment: a quick first stage — the device's smallest files, 1/32 of its bytes — in front of the largest-first list)
    for (size_t s = 0; s < ND; ++s) {
      std::vector<uint32_t>& f = dev_files[s];
      uint64_t acc = 0;
      size_t k = f.size();
      while (k > 1 && acc + bytes_of(f[k - 1]) <= dev_bytes[s] / 32) acc += bytes_of(f[--k]);
      std::rotate(f.begin(), f.begin() + k, f.end());
    }
  }
  // stage size per device: a quarter of its share, so that every device pipelines ≥ 4 stages (build ∥ prepare ∥ encode) however many devices split the list —
  // 26 input bytes per triangle is what pos + nrm + uv + indices come to; between 0.5M triangles (a stage pays fixed costs: its chain launch is bounded by
  // its longest stream) and 12M (one above ≈ 16M stops overlapping)
  for (size_t s = 0; s < ND; ++s) {
    R->devs.emplace_back(new PerDevice());
    PerDevice& d = *R->devs.back();
    d.self = R.get(); d.slot = (int)s;
    dmi_config c = cfg ? *cfg : dmi_config{};
    c.device = devs[s];
    const uint64_t expected = dev_bytes[s] / 26;
    const uint64_t stage = std::min<uint64_t>((uint64_t)12 << 20, std::max<uint64_t>((uint64_t)1 << 19, expected / 4));
    d.t = dmi_transcoder_create(&c, expected, stage, on_done, &d);
    if (!d.t) return host_fail(DMI_ERR_OUT_OF_MEMORY, "dmi_transcoder_create");
  }
  const unsigned host_thr = process_host_threads();
  // (when the last stage is coded all its files become ready at once and every other thread of the call is idle: half the host's threads write them)
  const unsigned n_assemblers = std::max(1u, std::min(8u, host_thr / 2));
  std::vector<dmi::Thread> assemblers;
  for (unsigned k = 0; k < n_assemblers; ++k) assemblers.emplace_back(assemble_loop, R.get());
  struct Stop { dmi_transcoded* R; std::vector<dmi::Thread>& th; ~Stop() { { std::lock_guard<std::mutex> lock(R->q_mutex); R->q_closed = true; } R->q_cv.notify_all(); for (auto& t : th) if (t.joinable()) t.join(); } } stop{R.get(), assemblers};

  R->assets.resize(n);
  // ---- parse: containers, JSON, plans, accessor descriptors — a pool of threads over the files in the order the devices will push them (round-robin over
  //      the devices' lists: every device's first files first) ----
  std::vector<uint32_t> parse_seq;
  parse_seq.reserve(n);
  for (size_t r = 0; parse_seq.size() < n; ++r) for (size_t s = 0; s < ND; ++s) if (r < dev_files[s].size()) parse_seq.push_back(dev_files[s][r]);
  std::unique_ptr<std::atomic<uint8_t>[]> state(new std::atomic<uint8_t>[n ? n : 1]);   // 0 pending, 1 parsed, 2 failed
  for (uint32_t i = 0; i < n; ++i) state[i].store(0, std::memory_order_relaxed);
  std::mutex parsed_mutex;
  std::condition_variable parsed_cv;
  std::atomic<bool> abort{false};
  std::atomic<uint32_t> next_parse{0}, in_place{0};
  std::atomic<uint64_t> parse_ns{0};
  std::atomic<double> last_parse_end{t_start};
  auto parse_one = [&](uint32_t i) -> int {
    R->assets[i].reset(new Asset());
    Asset& a = *R->assets[i];
    Span js;
    if (assets[i].glb) {
      Span bin;
      if (int rc = read_glb(assets[i].glb, assets[i].glb_bytes, &js, &bin)) return rc;
      a.buffers.push_back(bin);
    } else {
      js = Span{reinterpret_cast<const uint8_t*>(assets[i].json), assets[i].json_bytes};
      for (uint32_t b = 0; b < assets[i].n_buffers; ++b) a.buffers.push_back(Span{assets[i].buffers[b].data, assets[i].buffers[b].bytes});
    }
    std::string perr;
    if (!js.p || !json::parse(reinterpret_cast<const char*>(js.p), js.n, a.doc, perr) || !a.doc.is_object()) return gfail("asset " + std::to_string(i) + ": " + (perr.empty() ? "the JSON document is not an object" : perr));
    if (int rc = plan_asset(a)) return rc;
    a.raws.resize(a.prims.size());
    for (size_t k = 0; k < a.prims.size(); ++k) if (int rc = raw_of(a, a.prims[k], &a.raws[k])) return rc;
    uint32_t ip = 0;
    for (const Span& b : a.buffers) if (b.n && dmi_host_is_registered(b.p, b.n)) ++ip;
    if (ip) in_place.fetch_add(ip);
    return DMI_OK;
  };
  auto parse_loop = [&] {
    for (;;) {
      const uint32_t q = next_parse.fetch_add(1);
      if (q >= n || abort.load()) return;
      const uint32_t i = parse_seq[q];
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = parse_one(i);
      parse_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      if (rc) { R->fail_with(rc, dmi_last_error()); abort.store(true); }
      { std::lock_guard<std::mutex> lock(parsed_mutex); state[i].store(rc ? 2 : 1, std::memory_order_release); const double t = now_ms(); if (t > last_parse_end.load()) last_parse_end.store(t); }
      parsed_cv.notify_all();
    }
  };
  // ---- push: one pusher per device walks ITS list in order (a device's push order — and with it its stages — is the same from run to run) ----
  auto push_loop = [&](size_t s) {
    PerDevice& d = *R->devs[s];
    std::vector<dmi_raw_mesh> raws;
    for (uint32_t i : dev_files[s]) {
      {
        std::unique_lock<std::mutex> lock(parsed_mutex);
        parsed_cv.wait(lock, [&] { return state[i].load(std::memory_order_acquire) != 0 || abort.load(); });
      }
      if (abort.load() || state[i].load() != 1) return;
      Asset& a = *R->assets[i];
      const size_t np = a.prims.size();
      a.left.store((uint32_t)np);
      if (!np) { R->enqueue(i); continue; }   // (a file without a compressible primitive)
      // everything that touches the asset happens BEFORE its last primitive is pushed: once that push returns, an assembly thread may already be
      // releasing the asset's document and descriptors (the transcoder copied the descriptors it was handed)
      raws.assign(a.raws.begin(), a.raws.end());
      {
        std::lock_guard<std::mutex> lock(d.owner_mutex);
        for (size_t k = 0; k < np; ++k) {
          a.prims[k].device_slot = (int)s;
          a.prims[k].push_index = (uint32_t)d.owner.size();
          d.owner.emplace_back(i, (uint32_t)k);
          d.load += a.prims[k].triangles + 1;
          d.triangles += a.prims[k].triangles;
        }
        d.primitives += (uint32_t)np;
      }
      for (size_t k = 0; k < np; ++k) {
        if (int rc = dmi_transcoder_push(d.t, &raws[k], 1)) { R->fail_with(rc, dmi_last_error()); abort.store(true); parsed_cv.notify_all(); return; }
      }
    }
  };
  const unsigned n_parsers = std::max(1u, std::min({8u, host_thr / 4u, n ? n : 1u}));
  {
    std::vector<dmi::Thread> parsers, pushers;
    struct Join { std::vector<dmi::Thread>& a; std::vector<dmi::Thread>& b; ~Join() { for (auto& t : a) if (t.joinable()) t.join(); for (auto& t : b) if (t.joinable()) t.join(); } } join{parsers, pushers};
    for (unsigned k = 0; k < n_parsers; ++k) parsers.emplace_back(parse_loop);
    for (size_t s = 1; s < ND; ++s) pushers.emplace_back(push_loop, s);
    push_loop(0);   // (the caller's thread is the first device's pusher; the parsers run beside it)
  }
  const double t_pushed = now_ms();
  int rc = DMI_OK;
  std::string first_err;
  { std::lock_guard<std::mutex> lock(R->err_mutex); if (R->rc) { rc = R->rc; first_err = R->err; } }
  for (auto& d : R->devs) {
    const int r = dmi_transcoder_finish(d->t);
    if (r && !rc) { rc = r; first_err = dmi_last_error(); }
    double b = 0, p = 0, e = 0;
    (void)dmi_transcoder_timings(d->t, &b, &p, &e);
    R->stats.build_ms += b; R->stats.prepare_ms += p; R->stats.encode_ms += e;
    uint64_t nd = 0, nh = 0, np = 0;
    (void)dmi_transcoder_counts(d->t, &nd, &nh, &np);
    R->stats.primitives_device_built += (uint32_t)nd; R->stats.primitives_host_built += (uint32_t)nh; R->stats.primitives_in_place += (uint32_t)np;
    R->stats.stages += dmi_transcoder_stages(d->t);
    R->stats.triangles_in += d->triangles; R->stats.primitives += d->primitives;
  }
  const double t_finished = now_ms();
  { std::lock_guard<std::mutex> lock(R->q_mutex); R->q_closed = true; }
  R->q_cv.notify_all();
  for (auto& t : assemblers) t.join();
  assemblers.clear();
  if (!rc) { std::lock_guard<std::mutex> lock(R->err_mutex); if (R->rc) { rc = R->rc; first_err = R->err; } }
  if (rc) return host_fail(rc, first_err);
  const double t_assembled = now_ms();
  // the blobs are in the files now: the transcoders' buffers can go — on a pool thread (two thousand frees: 1.3 ms of a 67 ms call, nothing of it the caller's business)
  for (auto& d : R->devs) {
    dmi_transcoder* t = d->t;
    d->t = nullptr;
    try { pool_submit([t] { dmi_transcoder_destroy(t); }); } catch (...) { dmi_transcoder_destroy(t); }
  }
  if (dbg_on(DMI_DBG_TRACE | DMI_DBG_TRACE_STAGES))
    std::fprintf(stderr, "[dmi] transcode_assets: %u files on %zu device(s), %u parse threads: last file parsed %.1f ms, last push %.1f, last stage coded %.1f, files written %.1f, buffers released %.1f\n",
                 n, ND, n_parsers, last_parse_end.load() - t_start, t_pushed - t_start, t_finished - t_start, t_assembled - t_start, now_ms() - t_start);
  for (const auto& a : R->assets) R->stats.bytes_out += a->out_bytes;
  R->stats.bytes_in = in_bytes;
  R->stats.files = n;
  R->stats.buffers_in_place = in_place.load();
  R->stats.parse_ms = last_parse_end.load() - t_start;
  R->stats.parse_cpu_ms = (double)parse_ns.load() * 1e-6;
  R->stats.parse_threads = n_parsers;
  R->stats.pushed_ms = t_pushed - t_start; R->stats.finished_ms = t_finished - t_start;
  R->stats.assemble_ms = (double)R->assemble_ns.load() * 1e-6;
  R->stats.call_ms = now_ms() - t_start;
  R->stats.devices = (uint32_t)devs.size();
  *out = R.release();
  return DMI_OK;
}

int dmi_transcoded_file(const dmi_transcoded* r, uint32_t i, const uint8_t** glb, size_t* bytes, uint32_t* n_blobs) {
  if (!r || i >= r->assets.size()) return host_fail(DMI_ERR_INVALID_ARGUMENT, "file index out of range");
  const Asset& a = *r->assets[i];
  if (glb) *glb = a.out;
  if (bytes) *bytes = a.out_bytes;
  if (n_blobs) *n_blobs = (uint32_t)a.blobs.size();
  return DMI_OK;
}

int dmi_transcoded_blobs(const dmi_transcoded* r, uint32_t i, uint64_t* offsets, uint64_t* sizes) {
  if (!r || i >= r->assets.size()) return host_fail(DMI_ERR_INVALID_ARGUMENT, "file index out of range");
  const Asset& a = *r->assets[i];
  for (size_t k = 0; k < a.blobs.size(); ++k) { if (offsets) offsets[k] = a.blobs[k].first; if (sizes) sizes[k] = a.blobs[k].second; }
  return DMI_OK;
}

// every file and every blob of the result in two arrays (a caller in an interpreted language pays per call)
int dmi_transcoded_table(const dmi_transcoded* r, uint64_t* file_address, uint64_t* file_bytes, uint32_t* file_blobs, uint64_t* blob_offsets, uint64_t* blob_sizes, uint64_t blob_capacity) {
  if (!r) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  uint64_t at = 0;
  for (size_t i = 0; i < r->assets.size(); ++i) {
    const Asset& a = *r->assets[i];
    if (file_address) file_address[i] = (uint64_t)(uintptr_t)a.out;
    if (file_bytes) file_bytes[i] = a.out_bytes;
    if (file_blobs) file_blobs[i] = (uint32_t)a.blobs.size();
    for (const auto& b : a.blobs) {
      if (at < blob_capacity) { if (blob_offsets) blob_offsets[at] = b.first; if (blob_sizes) blob_sizes[at] = b.second; }
      ++at;
    }
  }
  return at > blob_capacity && (blob_offsets || blob_sizes) ? host_fail(DMI_ERR_INVALID_ARGUMENT, "blob table too small") : DMI_OK;
}

// the blocks of the output arena (every file lies inside one of them): a caller in an interpreted language wraps a few blocks once and slices its files
// out of them instead of wrapping a thousand files one by one.  Returns the number of blocks; fills at most `capacity` entries.
uint32_t dmi_transcoded_blocks(const dmi_transcoded* r, uint64_t* address, uint64_t* bytes, uint32_t capacity) {
  if (!r) return 0;
  const auto& bl = r->arena.blocks;
  for (size_t k = 0; k < bl.size() && k < capacity; ++k) { if (address) address[k] = (uint64_t)(uintptr_t)bl[k].p; if (bytes) bytes[k] = bl[k].cap; }
  return (uint32_t)bl.size();
}

int dmi_transcoded_stats(const dmi_transcoded* r, dmi_transcode_stats* s) {
  if (!r || !s) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  *s = r->stats;
  return DMI_OK;
}

// (the documents, descriptors and arena blocks of a call go back on a pool thread: a caller that frees one result as it takes the next — every loop over batches —
//  had ≈ 2 ms of destructors and munmap between the two)
void dmi_transcoded_free(dmi_transcoded* r) {
  if (!r) return;
  try { pool_submit([r] { delete r; }); } catch (...) { delete r; }
}

// the JSON layer on its own (host only): parse, write back compactly — what the tests pin against the interpreter's json module
int dmi_json_roundtrip(const char* text, size_t n, dmi_buffer* out) {
  if (!text || !out) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  Value v;
  std::string err, js;
  if (!json::parse(text, n, v, err)) return gfail(err);
  json::write(v, js);
  out->data = static_cast<uint8_t*>(std::malloc(js.size() ? js.size() : 1));
  if (!out->data) return host_fail(DMI_ERR_OUT_OF_MEMORY, "malloc");
  std::memcpy(out->data, js.data(), js.size());
  out->len = out->cap = js.size();
  return DMI_OK;
}

}  // extern "C"

namespace dmi { void gltf_pool_drop_all() { block_pool().drop_all(); } }
