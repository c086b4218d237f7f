// Tests of the C++ host side (include/draco_mi.hpp) written the way the reference writes its own:
//   core/mesh/builder.rs:405-440   test_with_tetrahedron
//   tests/compatibility.rs:7-17    en(): load a mesh, `encode(mesh, &mut writer, Config::default()).unwrap()`
// plus the error behaviour of the seam.  Expected bytes come from the CPU oracle (liboracle.so, test infrastructure), fed the same
// raw attribute rows through its own MeshBuilder restatement.
//   reference_style_tests --host    builder tests only (no GPU)
//   reference_style_tests           everything (needs an MI355X)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

#include "draco_mi.hpp"

extern "C" {
void* orc_session_new();
void orc_session_free(void*);
const char* orc_last_error();
void orc_builder_reset();
int orc_builder_add_attribute(const void* data, uint32_t count, int att_type, int domain, int comp_type, int ncomp, const uint32_t* parents, uint32_t nparents, int faithful);
void orc_builder_set_faces(const uint32_t* idx, uint32_t nfaces);
int orc_build(void* s, int faithful);
int orc_encode(void* s, const int* opts, int want_dump);
const uint8_t* orc_drc(void* s, uint64_t* len);
}

using namespace draco_oxide;
using core::AttributeDomain;
using core::AttributeType;
using core::MeshBuilder;
using core::NdVector;
using Vec3 = NdVector<float, 3>;
using Vec2 = NdVector<float, 2>;
using Faces = std::vector<std::array<size_t, 3>>;

static int g_failed = 0;
#define CHECK(cond, what)                                                                 \
  do { if (!(cond)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, what); ++g_failed; } } while (0)

// the oracle's bytes for the same builder input
struct RawMesh { std::vector<Vec3> pos; std::vector<Vec3> nrm; std::vector<Vec2> uv; Faces faces; };
static std::vector<uint8_t> oracle_encode(const RawMesh& m) {
  void* s = orc_session_new();
  orc_builder_reset();
  const uint32_t parent = 0;
  orc_builder_add_attribute(m.pos.data(), (uint32_t)m.pos.size(), DMI_ATT_POSITION, DMI_DOMAIN_POSITION, DMI_F32, 3, nullptr, 0, 0);
  if (!m.nrm.empty()) orc_builder_add_attribute(m.nrm.data(), (uint32_t)m.nrm.size(), DMI_ATT_NORMAL, DMI_DOMAIN_CORNER, DMI_F32, 3, &parent, 1, 0);
  if (!m.uv.empty()) orc_builder_add_attribute(m.uv.data(), (uint32_t)m.uv.size(), DMI_ATT_TEXCOORD, DMI_DOMAIN_CORNER, DMI_F32, 2, &parent, 1, 0);
  std::vector<uint32_t> idx;
  for (auto& f : m.faces) for (size_t v : f) idx.push_back((uint32_t)v);
  orc_builder_set_faces(idx.data(), (uint32_t)m.faces.size());
  std::vector<uint8_t> out;
  if (orc_build(s, 0) != 0 || orc_encode(s, nullptr, 0) != 0) { std::printf("  oracle: %s\n", orc_last_error()); orc_session_free(s); return out; }
  uint64_t n = 0;
  const uint8_t* p = orc_drc(s, &n);
  out.assign(p, p + n);
  orc_session_free(s);
  return out;
}
static core::Mesh build(const RawMesh& m) {
  auto builder = MeshBuilder::new_();
  builder.set_connectivity_attribute(m.faces);
  const auto pid = builder.add_attribute(m.pos, AttributeType::Position, AttributeDomain::Position, {});
  if (!m.nrm.empty()) builder.add_attribute(m.nrm, AttributeType::Normal, AttributeDomain::Corner, {pid});
  if (!m.uv.empty()) builder.add_attribute(m.uv, AttributeType::TextureCoordinate, AttributeDomain::Corner, {pid});
  return builder.build().expect("Failed to build mesh");
}

static RawMesh tetrahedron() {   // builder.rs:411-430: 12 corners, 4 distinct positions
  RawMesh m;
  m.faces = {{0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {9, 10, 11}};
  m.pos = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 0, 0}, {3, 0, 0}, {1, 0, 0}, {1, 0, 0}, {3, 0, 0}, {2, 0, 0}, {0, 0, 0}, {2, 0, 0}, {3, 0, 0}};
  return m;
}
// closed torus grid with analytic normals and a UV seam (u, v wrap: the last column / row duplicate positions with different UVs)
static RawMesh torus(size_t n) {
  RawMesh m;
  const double R = 1.0, r = 0.35, two_pi = 6.283185307179586;
  for (size_t i = 0; i <= n; ++i)
    for (size_t j = 0; j <= n; ++j) {
      const double u = (double)(i % n) / n, v = (double)(j % n) / n;
      const double cu = std::cos(two_pi * u), su = std::sin(two_pi * u), cv = std::cos(two_pi * v), sv = std::sin(two_pi * v);
      m.pos.push_back({(float)((R + r * cv) * cu), (float)((R + r * cv) * su), (float)(r * sv)});
      m.nrm.push_back({(float)(cv * cu), (float)(cv * su), (float)sv});
      m.uv.push_back({(float)i / n, (float)j / n});
    }
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < n; ++j) {
      const size_t a = i * (n + 1) + j, b = (i + 1) * (n + 1) + j, c = (i + 1) * (n + 1) + j + 1, d = i * (n + 1) + j + 1;
      m.faces.push_back({a, b, c});
      m.faces.push_back({a, c, d});
    }
  return m;
}
// OBJ → per-corner rows (both builders dedup them): v / vt / vn, polygons fan-triangulated
static RawMesh load_obj_rows(const std::string& path) {
  RawMesh m;
  std::vector<Vec3> v, vn;
  std::vector<Vec2> vt;
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "v") { Vec3 p; ss >> p[0] >> p[1] >> p[2]; v.push_back(p); }
    else if (tag == "vn") { Vec3 p; ss >> p[0] >> p[1] >> p[2]; vn.push_back(p); }
    else if (tag == "vt") { Vec2 p; ss >> p[0] >> p[1]; vt.push_back(p); }
    else if (tag == "f") {
      std::vector<size_t> corner;
      std::string tok;
      while (ss >> tok) {
        int iv = 0, it = 0, in_ = 0;
        std::sscanf(tok.c_str(), "%d/%d/%d", &iv, &it, &in_) == 3 || std::sscanf(tok.c_str(), "%d//%d", &iv, &in_) == 2 || std::sscanf(tok.c_str(), "%d/%d", &iv, &it) == 2 || std::sscanf(tok.c_str(), "%d", &iv);
        corner.push_back(m.pos.size());
        m.pos.push_back(v[(size_t)iv - 1]);
        if (in_) m.nrm.push_back(vn[(size_t)in_ - 1]);
        if (it) m.uv.push_back(vt[(size_t)it - 1]);
      }
      for (size_t k = 1; k + 1 < corner.size(); ++k) m.faces.push_back({corner[0], corner[k], corner[k + 1]});
    }
  }
  if (m.nrm.size() != m.pos.size()) m.nrm.clear();
  if (m.uv.size() != m.pos.size()) m.uv.clear();
  return m;
}

static void test_with_tetrahedron() {   // core/mesh/builder.rs:405-440
  std::printf("test_with_tetrahedron\n");
  const RawMesh t = tetrahedron();
  auto builder = MeshBuilder::new_();
  builder.set_connectivity_attribute(t.faces);
  builder.add_attribute(t.pos, AttributeType::Position, AttributeDomain::Position, {});
  auto mesh = builder.build().expect("Failed to build mesh");
  CHECK(mesh.get_faces().size() == 4, "Mesh should have 4 faces");
  CHECK(mesh.get_attributes().size() == 1, "Mesh should have 1 attribute");
  CHECK(mesh.get_attributes()[0].len() == 4, "Position attribute should have 4 vertices as duplicates are merged");
}
static void test_builder_rejects_a_mesh_without_positions() {   // builder.rs:115-125: the Position attribute is mandatory
  std::printf("test_builder_rejects_a_mesh_without_positions\n");
  auto builder = MeshBuilder::new_();
  builder.set_connectivity_attribute(Faces{{0, 1, 2}});
  builder.add_attribute(std::vector<Vec2>{{0, 0}, {1, 0}, {0, 1}}, AttributeType::TextureCoordinate, AttributeDomain::Corner, {});
  auto r = builder.build();
  CHECK(r.is_err(), "build() without a Position attribute is an Err");
  if (r.is_err()) { const Err e = r.unwrap_err(); CHECK(e.status != DMI_OK && !e.to_string().empty(), "the Err carries a status and a message"); }
}
static void test_position_is_moved_to_slot_zero() {   // builder.rs:115-125 get_sorted_attributes
  std::printf("test_position_is_moved_to_slot_zero\n");
  auto builder = MeshBuilder::new_();
  builder.set_connectivity_attribute(Faces{{0, 1, 2}, {2, 1, 3}});
  const auto uv = builder.add_attribute(std::vector<Vec2>{{0, 0}, {1, 0}, {0, 1}, {1, 1}}, AttributeType::TextureCoordinate, AttributeDomain::Corner, {1});
  const auto pos = builder.add_attribute(std::vector<Vec3>{{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}}, AttributeType::Position, AttributeDomain::Position, {});
  CHECK(uv == 0 && pos == 1, "ids are add-order indices");
  auto mesh = builder.build().expect("Failed to build mesh");
  CHECK(mesh.get_attributes()[0].get_attribute_type() == AttributeType::Position, "Position first");
  CHECK(mesh.get_attributes()[1].get_attribute_type() == AttributeType::TextureCoordinate, "then the texture coordinates");
}

static void en(const char* name, const RawMesh& raw) {   // tests/compatibility.rs:7-17
  std::printf("en(%s)\n", name);
  auto mesh = build(raw);
  std::vector<uint8_t> writer;
  encode::encode(std::move(mesh), writer, encode::Config::default_()).unwrap();
  const std::vector<uint8_t> want = oracle_encode(raw);
  CHECK(!want.empty(), "oracle produced a stream");
  CHECK(writer == want, "bit-exact with the reference algorithm (.drc bytes)");
  if (writer != want) std::printf("  %zu vs %zu bytes\n", writer.size(), want.size());
}
static void test_encode_appends_to_the_writer() {   // encode/mod.rs:59: `writer: &mut W`, never truncated
  std::printf("test_encode_appends_to_the_writer\n");
  const RawMesh raw = torus(9);
  std::vector<uint8_t> writer = {0xDE, 0xAD};
  encode::encode(build(raw), writer, encode::Config::default_()).unwrap();
  const std::vector<uint8_t> want = oracle_encode(raw);
  CHECK(writer.size() == want.size() + 2 && writer[0] == 0xDE && writer[1] == 0xAD && std::equal(want.begin(), want.end(), writer.begin() + 2), "previous bytes kept, stream appended");
}
static void test_the_two_halves_of_encode() {   // encode/mod.rs:83-93: header + connectivity, then the attribute section
  std::printf("test_the_two_halves_of_encode\n");
  const RawMesh raw = torus(17);
  auto mesh = build(raw);
  std::vector<uint8_t> writer;
  auto conn_out = encode::connectivity::encode_connectivity(mesh, writer).unwrap();
  const size_t head = writer.size();
  encode::attribute::encode_attributes(mesh, writer, conn_out, encode::Config::default_()).unwrap();
  CHECK(head > 11 && writer.size() > head, "both sections were appended");
  CHECK(writer == oracle_encode(raw), "connectivity + attributes == encode()");
}
static void test_zero_length_normal_is_an_err_not_an_abort() {   // prediction_transform/geom.rs:45 asserts in the reference
  std::printf("test_zero_length_normal_is_an_err_not_an_abort\n");
  RawMesh raw = torus(6);
  raw.uv.clear();
  raw.nrm[5] = {0, 0, 0};
  std::vector<uint8_t> writer;
  auto r = encode::encode(build(raw), writer, encode::Config::default_());
  CHECK(r.is_err() && writer.empty(), "Err, nothing written");
  if (r.is_err()) CHECK(r.unwrap_err().status == DMI_ERR_ZERO_NORMAL, "DMI_ERR_ZERO_NORMAL");
}

// io/gltf/encode.rs:1827-1842 — the transcoder's loop over a document's triangle primitives (MeshBuilder::build → encode::encode → bufferView), through
// the C ABI's dmi_transcoder: primitives pushed as raw accessors, stages built / prepared / encoded on library threads, results by push index.
struct DoneLog { std::vector<std::pair<uint32_t, uint32_t>> stages; };
static void on_stage_done(void* user, uint32_t first, uint32_t count) { static_cast<DoneLog*>(user)->stages.push_back({first, count}); }
static void test_transcoder_loop_over_primitives() {
  const std::vector<RawMesh> prims = {torus(9), tetrahedron(), torus(14), torus(6), torus(21)};
  std::vector<std::vector<uint32_t>> idx(prims.size());
  std::vector<std::vector<dmi_raw_accessor>> acc(prims.size());
  std::vector<dmi_raw_mesh> raw(prims.size());
  static const uint32_t parent0 = 0;
  uint64_t triangles = 0;
  for (size_t k = 0; k < prims.size(); ++k) {
    const RawMesh& m = prims[k];
    for (auto& f : m.faces) for (size_t v : f) idx[k].push_back((uint32_t)v);
    acc[k].push_back(dmi_raw_accessor{m.pos.data(), (uint32_t)m.pos.size(), 0, DMI_F32, 3, DMI_ATT_POSITION, DMI_DOMAIN_POSITION, 0, nullptr});
    if (!m.nrm.empty()) acc[k].push_back(dmi_raw_accessor{m.nrm.data(), (uint32_t)m.nrm.size(), 0, DMI_F32, 3, DMI_ATT_NORMAL, DMI_DOMAIN_CORNER, 1, &parent0});
    if (!m.uv.empty()) acc[k].push_back(dmi_raw_accessor{m.uv.data(), (uint32_t)m.uv.size(), 0, DMI_F32, 2, DMI_ATT_TEXCOORD, DMI_DOMAIN_CORNER, 1, &parent0});
    raw[k] = dmi_raw_mesh{acc[k].data(), (uint32_t)acc[k].size(), idx[k].data(), DMI_U32, (uint32_t)m.faces.size()};
    triangles += m.faces.size();
  }
  DoneLog log;
  dmi_transcoder* t = dmi_transcoder_create(nullptr, triangles, /*stage_triangles=*/300, on_stage_done, &log);
  CHECK(t != nullptr, "dmi_transcoder_create");
  CHECK(dmi_transcoder_reserve(t, (uint32_t)raw.size()) == DMI_OK, "reserve");
  CHECK(dmi_transcoder_push(t, raw.data(), 2) == DMI_OK && dmi_transcoder_push(t, raw.data() + 2, (uint32_t)raw.size() - 2) == DMI_OK, "push in two slices");
  CHECK(dmi_transcoder_finish(t) == DMI_OK, dmi_last_error());
  uint32_t seen = 0;
  for (auto& st : log.stages) seen += st.second;
  CHECK(seen == raw.size() && log.stages.size() > 1, "every primitive reported done, in more than one stage");
  for (size_t k = 0; k < prims.size(); ++k) {
    dmi_buffer head{}, section{};
    uint32_t nf = 0, np = 0;
    CHECK(dmi_transcoder_result(t, (uint32_t)k, &head, &section, &nf, &np) == DMI_OK, "result");
    std::vector<uint8_t> blob(head.data, head.data + head.len);
    blob.insert(blob.end(), section.data, section.data + section.len);
    const core::Mesh built = build(prims[k]);
    CHECK(nf == built.get_faces().size() && np == built.get_attributes()[0].len(), "face / point counts of the built mesh (the placeholder accessors)");
    CHECK(blob == oracle_encode(prims[k]), "blob == the reference algorithm's .drc of the same primitive");
  }
  dmi_transcoder_destroy(t);
}

int main(int argc, char** argv) {
  const bool host_only = argc > 1 && std::string(argv[1]) == "--host";
  const std::string data = argc > 2 ? argv[2] : "tests/golden/data";
  test_with_tetrahedron();
  test_builder_rejects_a_mesh_without_positions();
  test_position_is_moved_to_slot_zero();
  if (!host_only) {
    en("tetrahedron", tetrahedron());
    en("torus 24 (normals, UV seam)", torus(24));
    en("cube_quads.obj", load_obj_rows(data + "/cube_quads.obj"));
    en("sphere.obj", load_obj_rows(data + "/sphere.obj"));
    test_encode_appends_to_the_writer();
    test_the_two_halves_of_encode();
    test_zero_length_normal_is_an_err_not_an_abort();
    test_transcoder_loop_over_primitives();
  }
  std::printf(g_failed ? "%d check(s) FAILED\n" : "all checks passed\n", g_failed);
  return g_failed ? 1 : 0;
}
