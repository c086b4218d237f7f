// draco_mi.hpp — the host side of the boundary in C++ (header only, C++17), above the C ABI of draco_mi.h.
//
// The reference is a Rust crate; its toolchain is not part of this image, so the caller-facing layer a Rust user sees is mirrored
// here with the reference's own names, argument meaning, ownership and error behaviour (paths relative to draco-oxide/src/):
//
//   draco_oxide::core::MeshBuilder        core/mesh/builder.rs:15-90     new / add_attribute / set_connectivity_attribute / build
//   draco_oxide::core::Mesh, Attribute    core/mesh/mod.rs:13-23, core/attribute/mod.rs:26-49   get_faces / get_attributes / len …
//   draco_oxide::encode::Config           encode/mod.rs:32-42            Config::default()  (here: Config::default_(), `default` being a keyword)
//   draco_oxide::encode::encode           encode/mod.rs:59-97            encode(mesh, &mut writer, cfg) -> Result<(), Err>: the mesh is consumed,
//                                                                        bytes are APPENDED to the caller's writer
//   draco_oxide::Result<T, E>             Rust's Result: is_ok / is_err / unwrap / expect / unwrap_err (unwrap on an error throws)
//
// No arithmetic lives here: MeshBuilder::build is dmi_mesh_build (host), encode is dmi_encode_mesh (host connectivity + the MI355X
// attribute path).  Reference panics are error values (see dmi_status); nothing aborts across the boundary.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

#include "draco_mi.h"

namespace draco_oxide {

struct Unit {};   // Rust's ()

template <class T, class E>
class Result {
  std::variant<T, E> v_;
  explicit Result(std::variant<T, E> v) : v_(std::move(v)) {}

 public:
  static Result ok(T value) { return Result(std::variant<T, E>(std::in_place_index<0>, std::move(value))); }
  static Result err(E error) { return Result(std::variant<T, E>(std::in_place_index<1>, std::move(error))); }
  bool is_ok() const { return v_.index() == 0; }
  bool is_err() const { return v_.index() == 1; }
  T unwrap() {
    if (is_err()) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: " + std::get<1>(v_).to_string());
    return std::move(std::get<0>(v_));
  }
  T expect(const char* msg) {
    if (is_err()) throw std::runtime_error(std::string(msg) + ": " + std::get<1>(v_).to_string());
    return std::move(std::get<0>(v_));
  }
  E unwrap_err() {
    if (is_ok()) throw std::runtime_error("called `Result::unwrap_err()` on an `Ok` value");
    return std::move(std::get<1>(v_));
  }
};

// encode::Err (encode/mod.rs:44-55) / builder::Err: the stage enums of the reference collapse to the library's status + detail text
struct Err {
  int status = DMI_OK;    // dmi_status
  std::string detail;     // dmi_last_error() of the failing call
  std::string to_string() const { return std::string(dmi_strerror(status)) + " (" + detail + ")"; }
  static Err last(int status) { return Err{status, dmi_last_error()}; }
};

namespace core {

enum class AttributeType : uint8_t { Position = DMI_ATT_POSITION, Normal = DMI_ATT_NORMAL, Color = DMI_ATT_COLOR, TextureCoordinate = DMI_ATT_TEXCOORD,
                                     Custom = DMI_ATT_CUSTOM, Tangent = DMI_ATT_TANGENT, Material = DMI_ATT_MATERIAL, Joint = DMI_ATT_JOINT,
                                     Weight = DMI_ATT_WEIGHT };                                    // core/attribute/mod.rs:630-661
enum class AttributeDomain : uint8_t { Position = DMI_DOMAIN_POSITION, Corner = DMI_DOMAIN_CORNER };   // core/attribute/mod.rs:690-710
enum class ComponentDataType : uint8_t { U32 = DMI_U32, I32 = DMI_I32, F32 = DMI_F32 };               // the 4-byte types this path codes
using AttributeId = size_t;                                  // core/attribute/mod.rs: AttributeId (index in add order)
template <class T, size_t N> using NdVector = std::array<T, N>;   // core/shared: NdVector<N, T>

template <class T> constexpr ComponentDataType component_type_of() {
  static_assert(std::is_same_v<T, float> || std::is_same_v<T, uint32_t> || std::is_same_v<T, int32_t>, "attribute components must be f32, u32 or i32");
  return std::is_same_v<T, float> ? ComponentDataType::F32 : (std::is_same_v<T, uint32_t> ? ComponentDataType::U32 : ComponentDataType::I32);
}

// A built attribute: unique values in first-occurrence order + the point → value map (core/attribute/mod.rs:26-49).  A view into its Mesh.
class Attribute {
  const dmi_attribute* a_;
 public:
  explicit Attribute(const dmi_attribute* a) : a_(a) {}
  AttributeType get_attribute_type() const { return (AttributeType)a_->att_type; }
  AttributeDomain get_domain() const { return (AttributeDomain)a_->domain; }
  ComponentDataType get_component_type() const { return (ComponentDataType)a_->component_type; }
  size_t get_num_components() const { return a_->num_components; }
  size_t len() const { return a_->num_points; }                    // Attribute::len(): one entry per point
  size_t num_unique_values() const { return a_->num_unique; }
  AttributeId get_id() const { return a_->unique_id; }
  const void* unique_values() const { return a_->values; }
  const uint32_t* point_to_att_val_map() const { return a_->point_to_value; }   // nullptr = identity
  const dmi_attribute& raw() const { return *a_; }
};

// core/mesh/mod.rs:13-23.  Move-only (the Rust type is moved into encode()); owns the arrays MeshBuilder::build produced.
class Mesh {
  dmi_built_mesh m_{};
  friend class MeshBuilder;
 public:
  Mesh() = default;
  Mesh(const Mesh&) = delete;
  Mesh& operator=(const Mesh&) = delete;
  Mesh(Mesh&& o) noexcept : m_(o.m_) { o.m_ = dmi_built_mesh{}; }
  Mesh& operator=(Mesh&& o) noexcept { if (this != &o) { dmi_built_mesh_free(&m_); m_ = o.m_; o.m_ = dmi_built_mesh{}; } return *this; }
  ~Mesh() { dmi_built_mesh_free(&m_); }
  std::vector<std::array<size_t, 3>> get_faces() const {
    std::vector<std::array<size_t, 3>> f(m_.mesh.num_faces);
    for (uint32_t i = 0; i < m_.mesh.num_faces; ++i) f[i] = {m_.mesh.faces[3 * i], m_.mesh.faces[3 * i + 1], m_.mesh.faces[3 * i + 2]};
    return f;
  }
  std::vector<Attribute> get_attributes() const {
    std::vector<Attribute> a;
    for (uint32_t i = 0; i < m_.mesh.num_atts; ++i) a.emplace_back(&m_.mesh.atts[i]);
    return a;
  }
  const dmi_mesh& raw() const { return m_.mesh; }
};

// core/mesh/builder.rs:15-90
class MeshBuilder {
  struct Pending { std::vector<uint8_t> data; uint32_t num_points; uint8_t component_type, num_components, att_type, domain; std::vector<uint32_t> parents; };
  std::vector<Pending> atts_;
  std::vector<uint32_t> faces_;

 public:
  MeshBuilder() = default;
  static MeshBuilder new_() { return MeshBuilder(); }   // MeshBuilder::new()

  // builder.rs:30-39: one row per point; returns the attribute's id (its index in add order)
  template <class T, size_t N>
  AttributeId add_attribute(std::vector<NdVector<T, N>> data, AttributeType att_type, AttributeDomain domain, std::vector<AttributeId> parents) {
    static_assert(N >= 1 && N <= 4, "1 to 4 components");
    Pending p;
    p.num_points = (uint32_t)data.size();
    p.component_type = (uint8_t)component_type_of<T>();
    p.num_components = (uint8_t)N;
    p.att_type = (uint8_t)att_type;
    p.domain = (uint8_t)domain;
    p.data.resize(data.size() * sizeof(NdVector<T, N>));
    if (!data.empty()) std::memcpy(p.data.data(), data.data(), p.data.size());
    for (AttributeId id : parents) p.parents.push_back((uint32_t)id);
    atts_.push_back(std::move(p));
    return atts_.size() - 1;
  }
  // builder.rs:58-60
  void set_connectivity_attribute(std::vector<std::array<size_t, 3>> data) {
    faces_.clear();
    for (const auto& f : data) for (size_t v : f) faces_.push_back((uint32_t)v);
  }
  // builder.rs:62-90: dependency check, value dedup, Position to slot 0, point merge, degenerate faces and unreferenced points removed
  Result<Mesh, Err> build() {
    std::vector<dmi_raw_attribute> raw(atts_.size());
    for (size_t i = 0; i < atts_.size(); ++i) {
      const Pending& p = atts_[i];
      raw[i] = dmi_raw_attribute{p.data.data(), p.num_points, p.component_type, p.num_components, p.att_type, p.domain, (uint32_t)p.parents.size(),
                                 p.parents.empty() ? nullptr : p.parents.data()};
    }
    Mesh mesh;
    const int rc = dmi_mesh_build(raw.empty() ? nullptr : raw.data(), (uint32_t)raw.size(), faces_.empty() ? nullptr : faces_.data(), (uint32_t)(faces_.size() / 3), &mesh.m_);
    if (rc != DMI_OK) return Result<Mesh, Err>::err(Err::last(rc));
    return Result<Mesh, Err>::ok(std::move(mesh));
  }
};

}  // namespace core

namespace encode {

// encode/mod.rs:32-42 — only the default configuration is public in the reference; the quantization widths are what
// portabilization/mod.rs:118-134 derives from it (11 / 8 / 10 bits).  `device` / `stream` select where the attribute path runs.
struct Config {
  uint8_t position_quantization_bits = 11;
  uint8_t tex_coord_quantization_bits = 10;
  uint8_t generic_quantization_bits = 11;
  int device = 0;
  void* stream = nullptr;   // hipStream_t, nullptr = owned by the call
  static Config default_() { return Config{}; }   // Config::default()
  dmi_config raw() const {
    dmi_config c{};
    c.pos_bits = position_quantization_bits; c.uv_bits = tex_coord_quantization_bits; c.generic_bits = generic_quantization_bits;
    c.device = device; c.stream = stream;
    return c;
  }
};

using Err = draco_oxide::Err;

// encode/mod.rs:59-97: `encode(mesh, &mut writer, cfg)`.  The mesh is consumed; the `.drc` bytes (header, Edgebreaker connectivity,
// attribute section) are appended to `writer`, whose previous contents are left alone (the glTF path passes a fresh Vec: io/gltf/encode.rs:939).
template <class ByteWriter = std::vector<uint8_t>>
Result<Unit, Err> encode(core::Mesh mesh, ByteWriter& writer, const Config& cfg) {
  const dmi_config c = cfg.raw();
  dmi_buffer out{};
  const int rc = dmi_encode_mesh(&mesh.raw(), &c, &out);
  if (rc != DMI_OK) return Result<Unit, Err>::err(Err::last(rc));
  writer.insert(writer.end(), out.data, out.data + out.len);
  dmi_free(&out);
  return Result<Unit, Err>::ok(Unit{});
}

// The two halves of encode(), split at the seam a native replacement slots into (encode/mod.rs:86,90):
//   connectivity::encode_connectivity(faces, &mut atts, &mut writer, cfg)  -> ConnectivityEncoderOutput   (host: header is written by encode())
//   attribute::encode_attributes(atts, &mut writer, conn_out, &cfg)                                         (the MI355X hot path)
namespace connectivity {
// encode/connectivity/mod.rs:17-36.  Owns the flat corner tables, seeds (`corners_of_edgebreaker`) and sequences the attribute encoder consumes.
class ConnectivityEncoderOutput {
  dmi_conn c_{};
 public:
  ConnectivityEncoderOutput() = default;
  ConnectivityEncoderOutput(const ConnectivityEncoderOutput&) = delete;
  ConnectivityEncoderOutput& operator=(const ConnectivityEncoderOutput&) = delete;
  ConnectivityEncoderOutput(ConnectivityEncoderOutput&& o) noexcept : c_(o.c_) { o.c_ = dmi_conn{}; }
  ~ConnectivityEncoderOutput() { dmi_conn_free(&c_); }
  dmi_conn& raw() { return c_; }
  const dmi_conn& raw() const { return c_; }
};
// Appends the header (encode/header/mod.rs:26-54) and the Edgebreaker connectivity section to `writer`.
template <class ByteWriter = std::vector<uint8_t>>
Result<ConnectivityEncoderOutput, Err> encode_connectivity(const core::Mesh& mesh, ByteWriter& writer) {
  ConnectivityEncoderOutput out;
  dmi_buffer head{};
  const int rc = dmi_encode_connectivity(&mesh.raw(), &head, &out.raw());
  if (rc != DMI_OK) return Result<ConnectivityEncoderOutput, Err>::err(Err::last(rc));
  writer.insert(writer.end(), head.data, head.data + head.len);
  dmi_free(&head);
  return Result<ConnectivityEncoderOutput, Err>::ok(std::move(out));
}
}  // namespace connectivity

namespace attribute {
// encode/attribute/mod.rs:13-93: attribute i is coded against the universal corner table (i = 0) or attribute table i-1
// (all_inclusive_corner_table.rs:31-45); appends the attribute section to `writer`.
template <class ByteWriter = std::vector<uint8_t>>
Result<Unit, Err> encode_attributes(const core::Mesh& mesh, ByteWriter& writer, const connectivity::ConnectivityEncoderOutput& conn_out, const Config& cfg) {
  const dmi_config c = cfg.raw();
  const dmi_conn& k = conn_out.raw();
  dmi_buffer out{};
  const int rc = dmi_encode_attributes(mesh.raw().atts, k.tables, mesh.raw().num_atts, k.seeds, k.num_seeds, &c, &out);
  if (rc != DMI_OK) return Result<Unit, Err>::err(Err::last(rc));
  writer.insert(writer.end(), out.data, out.data + out.len);
  dmi_free(&out);
  return Result<Unit, Err>::ok(Unit{});
}
}  // namespace attribute

}  // namespace encode
}  // namespace draco_oxide
