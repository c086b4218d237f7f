// oracle/orc_attribute.cpp — TEST INFRASTRUCTURE (see oracle.hpp header).
// Restatement of the attribute-encoding hot path:
//   encode/attribute/{mod,attribute_encoder}.rs, encode/attribute/portabilization/*.rs,
//   encode/attribute/prediction_transform/{wrapped_difference,difference,oct_orthogonal,geom}.rs,
//   shared/attribute/prediction_scheme/{mesh_parallelogram_prediction,mesh_normal_prediction,
//   mesh_prediction_for_texture_coordinates,delta_prediction}.rs, encode/{mod,header/mod}.rs.
// Float arithmetic is IEEE f32 with each operation rounded separately: compile with
// -ffp-contract=off (see Makefile).  Integer arithmetic follows Rust release-mode semantics
// (wrapping), casts follow Rust `as` (float→int saturating, NaN→0) — quirk Q18.
#include <algorithm>
#include <chrono>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "oracle.hpp"

namespace orc {

namespace {

inline void trace(const char* what, size_t a = 0) { static const bool on = std::getenv("ORC_TRACE") != nullptr; if (on) std::fprintf(stderr, "[orc] %s %zu\n", what, a); }

inline i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
inline i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
inline i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
inline i64 wadd64(i64 a, i64 b) { return (i64)((u64)a + (u64)b); }
inline i64 wsub64(i64 a, i64 b) { return (i64)((u64)a - (u64)b); }
inline i64 wmul64(i64 a, i64 b) { return (i64)((u64)a * (u64)b); }
inline i64 wabs64(i64 a) { return a < 0 ? (i64)(0 - (u64)a) : a; }   // i64::abs wraps at MIN in release
inline i64 wdiv64(i64 a, i64 b) { if (a == std::numeric_limits<i64>::min() && b == -1) return a; return a / b; }

inline i32 f32_to_i32_sat(float f) {   // Rust `f as i32`
  if (f != f) return 0;
  if (f >= 2147483648.0f) return std::numeric_limits<i32>::max();
  if (f <= -2147483648.0f) return std::numeric_limits<i32>::min();
  return (i32)f;
}
inline i64 f32_to_i64_sat(float f) {   // Rust `f as i64`
  if (f != f) return 0;
  if (f >= 9223372036854775808.0f) return std::numeric_limits<i64>::max();
  if (f <= -9223372036854775808.0f) return std::numeric_limits<i64>::min();
  return (i64)f;
}
inline u8 zero_prob_u8(float p) {   // `(p as u16).clamp(1,255) as u8`
  u32 q;
  if (p != p || p <= 0.0f) q = 0; else if (p >= 65535.0f) q = 65535; else q = (u32)p;
  if (q < 1) q = 1;
  if (q > 255) q = 255;
  return (u8)q;
}

// A portabilized attribute: i32 AoS values (one row per *unique* value) + the original's p2v map.
struct PortAtt {
  u32 id = 0;
  AttType type = Position;
  int ncomp = 0;
  std::vector<i32> vals;
  bool has_map = false;
  const std::vector<u32>* p2v = nullptr;
  size_t num_unique = 0;
  size_t len() const { return has_map ? p2v->size() : num_unique; }
  const i32* get(u32 p) const { u32 vi = has_map ? (*p2v)[p] : p; return vals.data() + (size_t)vi * ncomp; }   // Attribute::get
};

// geom.rs:40-91 for f32 input (quirks Q5, Q6)
inline void octahedral_transform_f32(float x, float y, float z, float& ou, float& ov) {
  float abs_sum = std::fabs(x) + std::fabs(y) + std::fabs(z);
  float u = y / abs_sum;
  float v = z / abs_sum;
  if (x < 0.0f) {
    float u_out = (u < 0.0f) ? std::fabs(v) - 1.0f : 1.0f - std::fabs(v);
    float v_out = (v < 0.0f) ? std::fabs(u) - 1.0f : 1.0f - std::fabs(u);
    u = u_out;
    v = v_out;
  }
  ou = u;
  ov = v;
}

// geom.rs:137-157 (quirk Q7)
inline void into_faithful_oct_quantization(i32 u, i32 v, i32& ox, i32& oy) {
  const i32 max = 255, half = max / 2;
  i32 x = u, y = v;
  if ((u == 0 && v == 0) || (u == 255 && v == 0) || (u == 0 && v == 255)) { ox = 255; oy = 255; return; }
  else if (u == 0 && v > 127) y = half - (v - half);
  else if (u == max && v < half) y = half + (half - v);
  else if (v == max && u < half) x = half + (half - u);
  else if (v == 0 && u > half) x = half - (u - half);
  ox = x; oy = y;
}

// octahedral_quantization.rs:49-64 / mesh_normal_prediction.rs:120-127: (oct + 1) * 127 → trunc → faithful
inline void oct_quantize_f32(float x, float y, float z, i32& qx, i32& qy) {
  float u, v;
  octahedral_transform_f32(x, y, z, u, v);
  float a = (u + 1.0f) * 127.0f;
  float b = (v + 1.0f) * 127.0f;
  into_faithful_oct_quantization(f32_to_i32_sat(a), f32_to_i32_sat(b), qx, qy);
}

struct Ctx {
  const TableView& tv;
  const std::vector<u32>& seq;
  const PortAtt& att;
  const PortAtt* parent;
  bool faithful;
  std::vector<u32> record;   // vertices_processed_up_till_now
  std::vector<u32> rank;     // rank[v] = index in record (fast mode)
  bool contains(u32 v, size_t upto) const {
    if (faithful) { for (size_t k = 0; k < upto; ++k) if (record[k] == v) return true; return false; }
    return rank[v] != NONE && rank[v] < upto;
  }
};

// mesh_parallelogram_prediction.rs:186-237 (+ Q15)
void predict_parallelogram(const Ctx& cx, u32 c, size_t i, i32* out) {
  const int N = cx.att.ncomp;
  auto fallback = [&]() {
    if (i > 0) { const i32* v = cx.att.get(cx.tv.point_idx(cx.tv.left_most_corner(cx.record[i - 1]))); for (int k = 0; k < N; ++k) out[k] = v[k]; }
    else for (int k = 0; k < N; ++k) out[k] = 0;
  };
  u32 opp = cx.tv.opposite(c);
  if (opp == NONE) { fallback(); return; }
  u32 opp_v = cx.tv.vertex_idx(opp);
  u32 next_v = cx.tv.vertex_idx(TableView::next(c));
  u32 prev_v = cx.tv.vertex_idx(TableView::previous(c));
  if (!(cx.contains(opp_v, i) && cx.contains(next_v, i) && cx.contains(prev_v, i))) { fallback(); return; }
  const i32* a = cx.att.get(cx.tv.point_idx(TableView::next(c)));
  const i32* b = cx.att.get(cx.tv.point_idx(TableView::previous(c)));
  const i32* d = cx.att.get(cx.tv.point_idx(opp));
  for (int k = 0; k < N; ++k) out[k] = wsub(wadd(a[k], b[k]), d[k]);
}

// delta_prediction.rs:56-71
void predict_delta(const Ctx& cx, u32, size_t i, i32* out) {
  const int N = cx.att.ncomp;
  if (i == 0) { for (int k = 0; k < N; ++k) out[k] = 0; return; }
  const i32* v = cx.att.get(cx.tv.point_idx(cx.tv.left_most_corner(cx.record[i - 1])));
  for (int k = 0; k < N; ++k) out[k] = v[k];
}

// mesh_normal_prediction.rs:22-44,75-144
void predict_normal(const Ctx& cx, u32 c, std::vector<u8>& flips, i32* out) {
  const PortAtt& pos = *cx.parent;
  const i32* pc = pos.get(cx.tv.point_idx(c));
  const i32 pos_c[3] = {pc[0], pc[1], pc[2]};
  auto face_normal = [&](u32 cc, i64* acc) {
    const i32* pn = pos.get(cx.tv.point_idx(TableView::next(cc)));
    const i32* pp = pos.get(cx.tv.point_idx(TableView::previous(cc)));
    i32 dn[3], dp[3];
    for (int k = 0; k < 3; ++k) { dn[k] = wsub(pn[k], pos_c[k]); dp[k] = wsub(pp[k], pos_c[k]); }
    // cross in i32 (core/shared.rs:578-596), then widened
    i32 cr0 = wsub(wmul(dn[1], dp[2]), wmul(dn[2], dp[1]));
    i32 cr1 = wsub(wmul(dn[2], dp[0]), wmul(dn[0], dp[2]));
    i32 cr2 = wsub(wmul(dn[0], dp[1]), wmul(dn[1], dp[0]));
    acc[0] = wadd64(acc[0], cr0); acc[1] = wadd64(acc[1], cr1); acc[2] = wadd64(acc[2], cr2);
  };
  u32 curr = c;
  for (;;) { u32 l = cx.tv.swing_left(curr); if (l == NONE) break; curr = l; if (curr == c) break; }
  u32 start = curr;
  i64 sum[3] = {0, 0, 0};
  face_normal(curr, sum);
  for (;;) { u32 r = cx.tv.swing_right(curr); if (r == NONE) break; curr = r; if (curr == start) break; face_normal(curr, sum); }

  const i64 upper_bound = 1ll << 29;
  i64 abs_sum = wadd64(wadd64(wabs64(sum[0]), wabs64(sum[1])), wabs64(sum[2]));
  if (abs_sum > upper_bound) {
    i64 quotient = abs_sum / upper_bound;
    for (int k = 0; k < 3; ++k) sum[k] = wdiv64(sum[k], quotient);
  }
  i32 o3[3] = {(i32)sum[0], (i32)sum[1], (i32)sum[2]};
  i32 p0, p1;
  if (o3[0] == 0 && o3[1] == 0 && o3[2] == 0) { p0 = 0; p1 = 0; }
  else oct_quantize_f32((float)o3[0], (float)o3[1], (float)o3[2], p0, p1);   // geom.rs:46-55: i32 → f32 first
  const i32* actual = cx.att.get(cx.tv.point_idx(c));
  i32 d10 = wsub(p0, actual[0]), d11 = wsub(p1, actual[1]);
  i32 n0 = wmul(p0, -1), n1 = wmul(p1, -1);
  i32 d20 = wsub(n0, actual[0]), d21 = wsub(n1, actual[1]);
  i32 dot1 = wadd(wmul(d10, d10), wmul(d11, d11));
  i32 dot2 = wadd(wmul(d20, d20), wmul(d21, d21));
  if (dot1 > dot2) { flips.push_back(1); p0 = n0; p1 = n1; } else flips.push_back(0);
  out[0] = p0; out[1] = p1;
}

// mesh_prediction_for_texture_coordinates.rs:32-48
u64 int_sqrt(u64 value) {
  if (value == 0) return 0;
  u64 act = value, sq = 1;
  while (act >= 2) { sq *= 2; act /= 4; }
  sq = (sq + value / sq) / 2;
  while (sq * sq > value) sq = (sq + value / sq) / 2;
  return sq;
}

// mesh_prediction_for_texture_coordinates.rs:51-81
void texcoord_fallback(const Ctx& cx, u32 c, size_t i, i32* out) {
  u32 next_corner = TableView::next(c);
  u32 next_vertex = cx.tv.vertex_idx(next_corner);
  if (cx.contains(next_vertex, i)) { const i32* v = cx.att.get(cx.tv.point_idx(next_corner)); out[0] = v[0]; out[1] = v[1]; return; }
  if (i > 0) { const i32* v = cx.att.get(cx.tv.point_idx(cx.tv.left_most_corner(cx.record[i - 1]))); out[0] = v[0]; out[1] = v[1]; return; }
  out[0] = 0; out[1] = 0;
}

// mesh_prediction_for_texture_coordinates.rs:107-219
void predict_texcoord(const Ctx& cx, u32 c, size_t i, std::vector<u8>& orientation, i32* out) {
  const PortAtt& pos = *cx.parent;
  u32 next_corner = TableView::next(c), prev_corner = TableView::previous(c);
  u32 next_pt = cx.tv.point_idx(next_corner), prev_pt = cx.tv.point_idx(prev_corner), curr_pt = cx.tv.point_idx(c);
  u32 next_vertex = cx.tv.vertex_idx(next_corner), prev_vertex = cx.tv.vertex_idx(prev_corner);
  if (cx.contains(next_vertex, i) && cx.contains(prev_vertex, i)) {
    const i32* cu = cx.att.get(curr_pt); const i32* nu = cx.att.get(next_pt); const i32* pu = cx.att.get(prev_pt);
    i64 curr_uv[2] = {cu[0], cu[1]}, next_uv[2] = {nu[0], nu[1]}, prev_uv[2] = {pu[0], pu[1]};
    if (next_uv[0] == prev_uv[0] && next_uv[1] == prev_uv[1]) { out[0] = pu[0]; out[1] = pu[1]; return; }
    auto getpos = [&](u32 p, i64* o) {   // get_position_for_vertex :22-30
      if ((size_t)p < pos.len()) { const i32* v = pos.get(p); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; } else { o[0] = o[1] = o[2] = 0; }
    };
    i64 curr_pos[3], next_pos[3], prev_pos[3];
    getpos(curr_pt, curr_pos); getpos(next_pt, next_pos); getpos(prev_pt, prev_pos);
    i64 pn[3];
    for (int k = 0; k < 3; ++k) pn[k] = wsub64(prev_pos[k], next_pos[k]);
    u64 pn_norm2_squared = (u64)wadd64(wadd64(wmul64(pn[0], pn[0]), wmul64(pn[1], pn[1])), wmul64(pn[2], pn[2]));
    if (pn_norm2_squared != 0) {
      i64 cn[3];
      for (int k = 0; k < 3; ++k) cn[k] = wsub64(curr_pos[k], next_pos[k]);
      i64 cn_dot_pn = wadd64(wadd64(wmul64(pn[0], cn[0]), wmul64(pn[1], cn[1])), wmul64(pn[2], cn[2]));
      i64 pn_uv[2] = {wsub64(prev_uv[0], next_uv[0]), wsub64(prev_uv[1], next_uv[1])};
      const i64 I64MAX = std::numeric_limits<i64>::max();
      i64 n_uv_absmax = std::max(wabs64(next_uv[0]), wabs64(next_uv[1]));
      // `i64::MAX / pn_norm2_squared as i64`: a norm² ≥ 2^63 would cast negative; quantised inputs never reach it.
      if (n_uv_absmax > wdiv64(I64MAX, (i64)pn_norm2_squared)) { texcoord_fallback(cx, c, i, out); return; }
      i64 pn_uv_absmax = std::max(wabs64(pn_uv[0]), wabs64(pn_uv[1]));
      if (wabs64(cn_dot_pn) > wdiv64(I64MAX, pn_uv_absmax)) { texcoord_fallback(cx, c, i, out); return; }
      i64 x_uv[2];
      for (int k = 0; k < 2; ++k) x_uv[k] = wadd64(wmul64(next_uv[k], (i64)pn_norm2_squared), wmul64(pn_uv[k], cn_dot_pn));
      i64 pn_absmax = std::max(std::max(wabs64(pn[0]), wabs64(pn[1])), wabs64(pn[2]));
      if (wabs64(cn_dot_pn) > wdiv64(I64MAX, pn_absmax)) { texcoord_fallback(cx, c, i, out); return; }
      i64 x_pos[3];
      for (int k = 0; k < 3; ++k) x_pos[k] = wadd64(next_pos[k], wdiv64(wmul64(pn[k], cn_dot_pn), (i64)pn_norm2_squared));
      i64 cxv[3];
      for (int k = 0; k < 3; ++k) cxv[k] = wsub64(curr_pos[k], x_pos[k]);
      u64 cx_norm2_squared = (u64)wadd64(wadd64(wmul64(cxv[0], cxv[0]), wmul64(cxv[1], cxv[1])), wmul64(cxv[2], cxv[2]));
      i64 cx_uv[2] = {pn_uv[1], (i64)(0 - (u64)pn_uv[0])};
      u64 norm_squared = int_sqrt(cx_norm2_squared * pn_norm2_squared);
      cx_uv[0] = wmul64(cx_uv[0], (i64)norm_squared);
      cx_uv[1] = wmul64(cx_uv[1], (i64)norm_squared);
      i64 p0[2], p1[2];
      for (int k = 0; k < 2; ++k) {
        p0[k] = wdiv64(wadd64(x_uv[k], cx_uv[k]), (i64)pn_norm2_squared);
        p1[k] = wdiv64(wsub64(x_uv[k], cx_uv[k]), (i64)pn_norm2_squared);
      }
      i64 e0[2] = {wsub64(curr_uv[0], p0[0]), wsub64(curr_uv[1], p0[1])};
      i64 e1[2] = {wsub64(curr_uv[0], p1[0]), wsub64(curr_uv[1], p1[1])};
      i64 dist0 = wadd64(wmul64(e0[0], e0[0]), wmul64(e0[1], e0[1]));
      i64 dist1 = wadd64(wmul64(e1[0], e1[0]), wmul64(e1[1], e1[1]));
      if (dist0 < dist1) { orientation.push_back(1); out[0] = (i32)p0[0]; out[1] = (i32)p0[1]; }
      else { orientation.push_back(0); out[0] = (i32)p1[0]; out[1] = (i32)p1[1]; }
      return;
    }
  }
  texcoord_fallback(cx, c, i, out);
}

}  // namespace

// oct_orthogonal.rs:23-74 (a10)
void oct_orthogonal_map(const i32* orig_in, const i32* pred_in, i32* corr) {
  const i32 one = 255 / 2;
  i32 p0 = wsub(pred_in[0], one), p1 = wsub(pred_in[1], one);
  i32 o0 = wsub(orig_in[0], one), o1 = wsub(orig_in[1], one);
  auto sgn = [](i32 v) -> i32 { return v > 0 ? 1 : (v < 0 ? -1 : 0); };
  auto iabs = [](i32 v) -> i32 { return v < 0 ? (i32)(0u - (u32)v) : v; };
  if (wadd(iabs(p0), iabs(p1)) > one) {
    i32 pred0 = p0;
    i32 qs = -sgn(wmul(p0, p1));
    p0 = wadd(wmul(qs, p1), wmul(sgn(p0), one));
    p1 = wadd(wmul(qs, pred0), wmul(sgn(p1), one));
    i32 orig0 = o0;
    i32 qo = -sgn(wmul(o0, o1));
    o0 = wadd(wmul(qo, o1), wmul(sgn(o0), one));
    o1 = wadd(wmul(qo, orig0), wmul(sgn(o1), one));
  }
  if (!(p0 == 0 && p1 == 0)) {
    while (p0 >= 0 || p1 > 0) {
      i32 t = p0; p0 = (i32)(0u - (u32)p1); p1 = t;
      t = o0; o0 = (i32)(0u - (u32)o1); o1 = t;
    }
  }
  i32 c0 = wsub(o0, p0), c1 = wsub(o1, p1);
  if (c0 < 0) c0 = wadd(c0, 255);
  if (c1 < 0) c1 = wadd(c1, 255);
  corr[0] = c0; corr[1] = c1;
}

namespace {

enum Scheme : u8 { SchDelta = 0, SchParallelogram = 1, SchTexCoord = 5, SchNormal = 6 };   // prediction_scheme/mod.rs:74-86
enum Transform : u8 { TrDifference = 0, TrWrapped = 1, TrOctOrth = 3 };                  // prediction_transform/mod.rs:92-101
enum PortType : u8 { PortToBits = 1, PortCoordwise = 2, PortOct = 3 };                   // portabilization/mod.rs:85-92

PortType port_type_for(AttType t) { return t == Normal ? PortOct : (t == Custom ? PortToBits : PortCoordwise); }   // :102-108

}  // namespace

double g_stage_seconds[kStageSlots] = {0};   // see oracle.hpp
double g_rans_symbols = 0;

// encode/attribute/mod.rs:13-93
std::string encode_attributes(const std::vector<Attribute>& atts, const ConnOutput& conn, const Options& opt, Bytes& w, Blobs* dump) {
  w.w8((u8)atts.size());                                         // :26
  for (size_t i = 0; i < atts.size(); ++i) {                     // :30-39
    w.w8((u8)((u8)i - 1));                                       // wrapping_sub (Q13)
    w.w8((u8)atts[i].domain);
    w.w8(0);                                                     // TraversalType::DepthFirst
  }
  for (auto& a : atts) {                                         // :43-57
    w.w8(1);
    w.w8((u8)a.type);
    w.w8((u8)a.ctype);
    w.w8((u8)a.ncomp);
    w.w8(0);
    w.w8((u8)a.id);
    w.w8((u8)port_type_for(a.type));
  }
  std::vector<PortAtt> port_atts;
  port_atts.reserve(atts.size());
  for (size_t i = 0; i < atts.size(); ++i) {
    const Attribute& att = atts[i];
    const std::string tag = "att" + std::to_string(i);
    // parents: looked up among already-portabilized attributes (:63-66, Q17)
    std::vector<const PortAtt*> parents;
    for (u32 pid : att.parents) {
      const PortAtt* found = nullptr;
      for (auto& p : port_atts) if (p.id == pid) { found = &p; break; }
      if (!found) return "parent attribute not encoded yet (reference unwrap() panic, encode/attribute/mod.rs:65)";
      parents.push_back(found);
    }
    // Config::default_for, attribute_encoder.rs:59-108
    Scheme scheme; Transform transform;
    switch (att.type) {
      case Position: scheme = SchParallelogram; transform = TrWrapped; break;
      case Normal: scheme = SchNormal; transform = TrOctOrth; break;
      case TextureCoordinate: scheme = SchTexCoord; transform = TrWrapped; break;
      case Custom: scheme = SchParallelogram; transform = TrWrapped; break;
      default: scheme = SchDelta; transform = TrDifference; break;
    }
    if (att.type == Position && opt.positions_delta) { scheme = SchDelta; transform = TrDifference; }   // internal-config variant
    const size_t att_begin = w.size();
    w.w8((u8)scheme);      // attribute_encoder.rs:159
    w.w8((u8)transform);   // :160

    // encode_typed :229-272 — table choice (all_inclusive_corner_table.rs:31-45)
    TableView tv;
    tv.ct = &conn.ct;
    tv.at = (i > 0 && i - 1 < conn.att_tables.size()) ? &conn.att_tables[i - 1] : nullptr;
    trace("attribute", i);
    const auto tseq0 = std::chrono::steady_clock::now();
    std::vector<u32> seq = compute_sequence(tv, conn.corners_of_edgebreaker, opt.faithful);
    g_stage_seconds[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - tseq0).count();
    trace("sequence", seq.size());

    // Portabilization::new + portabilize, :283-299
    Bytes port_info;
    PortAtt pa;
    std::unique_ptr<StageTimer> t_quant(new StageTimer(4));
    pa.id = att.id; pa.type = att.type; pa.has_map = att.has_map; pa.p2v = &att.p2v; pa.num_unique = att.num_unique();
    const PortType pt = port_type_for(att.type);
    if (pt == PortToBits) {
      if (comp_size(att.ctype) != 4) return "ToBits on a non-4-byte component type (reference size assert, core/buffer/attribute.rs:54-58)";
      pa.ncomp = att.ncomp;
      pa.vals.resize(att.num_unique() * att.ncomp);
      if (!pa.vals.empty()) std::memcpy(pa.vals.data(), att.data.data(), pa.vals.size() * 4);
    } else {
      if (att.ctype != F32) return "UnsupportedDataType (oracle restates the f32 path only)";
      const int N = att.ncomp;
      if (N < 1 || N > 4) return "UnsupportedNumComponents";
      const size_t nu = att.num_unique();
      if (pt == PortCoordwise) {
        // quantization_coordinate_wise.rs:24-68 (Q1, Q2)
        int bits = att.type == Position ? opt.pos_bits : (att.type == TextureCoordinate ? opt.uv_bits : opt.generic_bits);
        float mn[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
        for (size_t v = 0; v < nu; ++v) { const float* p = att.f32_at((u32)v); for (int k = 0; k < N; ++k) if (p[k] < mn[k]) mn[k] = p[k]; }
        for (size_t v = 0; v < nu; ++v) { const float* p = att.f32_at((u32)v); for (int k = 0; k < N; ++k) if (p[k] > mx[k]) mx[k] = p[k]; }
        float delta_max = 0.0f;
        for (int k = 0; k < N; ++k) { float d = mx[k] - mn[k]; if (d > delta_max) delta_max = d; }
        for (int k = 0; k < N; ++k) port_info.wf32(mn[k]);
        port_info.wf32(delta_max);
        port_info.w8((u8)bits);
        // portabilize_value :70-91 (Q3)
        const float maxq = (float)(u64)((1ull << bits) - 1);
        pa.ncomp = N;
        pa.vals.resize(nu * N);
        for (size_t v = 0; v < nu; ++v) {
          const float* p = att.f32_at((u32)v);
          for (int k = 0; k < N; ++k) {
            float diff = p[k] - mn[k];
            float normalized = (delta_max == 0.0f) ? diff : diff / delta_max;
            float quantized = normalized * maxq;
            pa.vals[v * N + k] = (i32)f32_to_i64_sat(quantized + 0.5f);
          }
        }
      } else {
        // octahedral_quantization.rs:33-64
        if (N != 3) return "octahedral quantization needs 3 components (geom.rs:44 assert)";
        port_info.w8(8);
        pa.ncomp = 2;
        pa.vals.resize(nu * 2);
        for (size_t v = 0; v < nu; ++v) {
          const float* p = att.f32_at((u32)v);
          if (p[0] == 0.0f && p[1] == 0.0f && p[2] == 0.0f) return "zero normal (reference assert, geom.rs:45)";
          oct_quantize_f32(p[0], p[1], p[2], pa.vals[v * 2], pa.vals[v * 2 + 1]);
        }
      }
    }
    t_quant.reset();
    if (pa.ncomp < 1 || pa.ncomp > 4) return "UnsupportedNumComponents";
    const int N = pa.ncomp;

    // encode_portabilized :312-389
    const PortAtt* parent = nullptr;
    if (scheme == SchNormal) {
      if (parents.size() != 1) return "MeshNormalPrediction requires exactly one parent (mesh_normal_prediction.rs:57)";
      if (parents[0]->type != Position) return "MeshNormalPrediction requires a Position parent (mesh_normal_prediction.rs:58-61)";
      if (N != 2) return "normal attribute must portabilize to 2 components";
      parent = parents[0];
    } else if (scheme == SchTexCoord) {
      if (parents.empty()) return "texture-coordinate prediction needs parents[0] (reference index panic)";
      if (parents[0]->ncomp != 3) return "texture-coordinate parent must have 3 components";
      if (N != 2) return "texture coordinates must have 2 components";
      parent = parents[0];
    }
    if (transform == TrOctOrth && N != 2) return "oct transform needs N == 2 (oct_orthogonal.rs:29)";
    std::unique_ptr<StageTimer> t_pred(new StageTimer(5));
    Ctx cx{tv, seq, pa, parent, opt.faithful, {}, {}};
    cx.record.reserve(seq.size());
    if (!opt.faithful) {
      cx.rank.assign(tv.num_vertices(), NONE);
      for (size_t k = 0; k < seq.size(); ++k) cx.rank[tv.vertex_idx(seq[k])] = (u32)k;
      // `rank[v] < i` ⇔ vertices_up_till_now.contains(v): every vertex is emitted once (sequence.rs:41-46)
    }
    trace("portabilized", pa.vals.size());
    std::vector<i32> origs(seq.size() * N), preds(seq.size() * N);
    std::vector<u8> flips, orientation;
    for (size_t k = 0; k < seq.size(); ++k) {   // :332-338
      u32 c = seq[k];
      i32* pr = preds.data() + k * N;
      switch (scheme) {
        case SchParallelogram: predict_parallelogram(cx, c, k, pr); break;
        case SchDelta: predict_delta(cx, c, k, pr); break;
        case SchNormal: predict_normal(cx, c, flips, pr); break;
        case SchTexCoord: predict_texcoord(cx, c, k, orientation, pr); break;
      }
      cx.record.push_back(tv.vertex_idx(c));
      const i32* o = pa.get(tv.point_idx(c));
      for (int j = 0; j < N; ++j) origs[k * N + j] = o[j];
    }
    trace("predicted", flips.size() + orientation.size());
    t_pred.reset();
    std::unique_ptr<StageTimer> t_tr(new StageTimer(6));
    // transform: map_with_tentative_metadata + squeeze
    Bytes transform_info;
    std::vector<u32> symbols(seq.size() * N);
    if (transform == TrWrapped) {   // wrapped_difference.rs:36-99 (Q16)
      i32 mx = std::numeric_limits<i32>::min(), mn = std::numeric_limits<i32>::max();
      for (i32 v : origs) { if (v > mx) mx = v; if (v < mn) mn = v; }
      i32 diff = wsub(mx, mn);
      i32 max_diff = wadd(1, diff);
      i32 max_corr = max_diff / 2;
      i32 min_corr = (i32)(0u - (u32)max_corr);
      if ((max_diff & 1) == 0) max_corr = wsub(max_corr, 1);
      for (size_t k = 0; k < origs.size(); ++k) {
        i32 p = preds[k];
        if (mn <= mx) p = p < mn ? mn : (p > mx ? mx : p);   // Ord::clamp (asserts min<=max; only violated when empty)
        i32 val = wsub(origs[k], p);
        i32 corr;
        if (val > max_corr) corr = wsub(val, max_diff);
        else if (val < min_corr) corr = wadd(val, max_diff);
        else corr = val;
        symbols[k] = (u32)to_positive_i32(corr);
      }
      transform_info.wi32(mn);
      transform_info.wi32(mx);
    } else if (transform == TrDifference) {   // difference.rs:26-34
      for (size_t k = 0; k < origs.size(); ++k) symbols[k] = (u32)to_positive_i32(wsub(origs[k], preds[k]));
    } else {   // oct_orthogonal.rs
      for (size_t k = 0; k < seq.size(); ++k) { i32 corr[2]; oct_orthogonal_map(&origs[k * 2], &preds[k * 2], corr); symbols[k * 2] = (u32)corr[0]; symbols[k * 2 + 1] = (u32)corr[1]; }
      transform_info.w32(255);
      transform_info.w32(255 / 2);
    }
    t_tr.reset();
    w.w8(1);   // rans_encoding flag :344
    // symbols are cast `as u64` from i32 (:347-350): a negative i32 would sign-extend to a huge
    // index; the transforms above only produce non-negatives for sane inputs.
    for (u32 s : symbols) if ((i32)s < 0) return "negative symbol (reference would index out of bounds)";
    { u32 mxs = 0; for (u32 s2 : symbols) if (s2 > mxs) mxs = s2; trace("max symbol", mxs); }
    std::string e = encode_symbols_direct(symbols, w);
    trace("symbols coded", w.size());
    if (!e.empty()) return e;

    auto write_rabs_block = [&](u8 zero_prob, const std::vector<u8>& bits) -> std::string {
      StageTimer t_rabs(8);
      RabsCoder rc(zero_prob);
      for (u8 b : bits) rc.write(b);
      Bytes b;
      std::string er = rc.flush(b);
      if (!er.empty()) return er;
      leb128_write(b.size(), w);
      w.append(b);
      return "";
    };
    if (scheme == SchNormal) {   // :362-366 + mesh_normal_prediction.rs:147-163 (Q9)
      w.append(transform_info);
      size_t c0 = 0;
      for (u8 f : flips) if (!f) ++c0;
      u8 zp = zero_prob_u8(((float)c0 / (float)flips.size()) * 256.0f + 0.5f);
      w.w8(zp);
      e = write_rabs_block(zp, flips);
      if (!e.empty()) return e;
    } else if (scheme == SchTexCoord) {   // :367-371 + mesh_prediction_for_texture_coordinates.rs:221-260 (Q10)
      size_t c0 = 0;
      { bool last = true; for (u8 o : orientation) { bool ob = o != 0; if (ob == last) continue; last = ob; ++c0; } }
      float len_f = (float)orientation.size() + 0.001f;
      u8 zp = zero_prob_u8(((float)c0 / len_f) * 256.0f + 0.5f);
      w.w32((u32)orientation.size());
      w.w8(zp);
      std::vector<u8> bits(orientation.size());
      { bool last = true; for (size_t k = orientation.size(); k-- > 0;) { bool ob = orientation[k] != 0; if (ob == last) bits[k] = 1; else { last = ob; bits[k] = 0; } } }
      e = write_rabs_block(zp, bits);
      if (!e.empty()) return e;
      w.append(transform_info);
    } else {
      w.append(transform_info);   // :372-382
    }
    w.append(port_info);   // :384-386

    if (dump) {
      blob_put(dump, tag + ".seq", seq);
      blob_put(dump, tag + ".q", pa.vals);
      blob_put(dump, tag + ".sym", symbols);
      blob_put(dump, tag + ".pred", preds);
      blob_put(dump, tag + ".flips", flips);
      blob_put(dump, tag + ".orient", orientation);
      std::vector<u8> blk(w.begin() + att_begin, w.end());
      (*dump)[tag + ".bytes"] = blk;
    }
    port_atts.push_back(std::move(pa));
  }
  return "";
}

// encode/mod.rs:59-97 + encode/header/mod.rs:26-54
std::string encode_mesh(const Mesh& mesh, const Options& opt, Bytes& w, Blobs* dump) {
  for (char ch : std::string("DRACO")) w.w8((u8)ch);
  w.w8(2); w.w8(2);
  w.w8(1);      // EncodedGeometryType::TrianglarMesh
  w.w8(1);      // EncoderMethod::Edgebreaker
  w.w16(0);     // flags (metadata off)
  const size_t conn_begin = w.size();
  ConnOutput conn;
  trace("connectivity begin");
  for (double& v : g_stage_seconds) v = 0.0;
  g_rans_symbols = 0;
  const auto t0 = std::chrono::steady_clock::now();
  std::string e = encode_connectivity(mesh, w, conn);
  if (!e.empty()) return e;
  const auto t1 = std::chrono::steady_clock::now();
  g_stage_seconds[0] = std::chrono::duration<double>(t1 - t0).count();
  trace("connectivity bytes", w.size());
  const size_t att_begin = w.size();
  e = encode_attributes(mesh.atts, conn, opt, w, dump);
  if (!e.empty()) return e;
  g_stage_seconds[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  if (dump) {
    (*dump)["conn.bytes"] = std::vector<u8>(w.begin() + conn_begin, w.begin() + att_begin);
    (*dump)["atts.bytes"] = std::vector<u8>(w.begin() + att_begin, w.end());
    blob_put(dump, "conn.corners", conn.corners_of_edgebreaker);
    (*dump)["conn.symbols"] = std::vector<u8>(conn.symbols.begin(), conn.symbols.end());
    blob_put(dump, "ct.opp", conn.ct.opposite_corners);
    blob_put(dump, "ct.lmc", conn.ct.left_most_corners);
    std::vector<u32> c2v(conn.ct.num_corners()), c2p(conn.ct.num_corners());
    for (u32 c = 0; c < conn.ct.num_corners(); ++c) { c2v[c] = conn.ct.vertex_idx(c); c2p[c] = conn.ct.point_idx(c); }
    blob_put(dump, "ct.c2v", c2v);
    blob_put(dump, "ct.c2p", c2p);
    std::vector<u32> nv{conn.ct.num_vertices()};
    blob_put(dump, "ct.nverts", nv);
    for (size_t j = 0; j < conn.att_tables.size(); ++j) {
      const auto& t = conn.att_tables[j];
      const std::string tag = "at" + std::to_string(j);
      blob_put(dump, tag + ".c2v", t.corner_to_vertex);
      blob_put(dump, tag + ".lmc", t.left_most_corners);
      blob_put(dump, tag + ".seam", t.is_edge_on_seam);
      blob_put(dump, tag + ".v2a", t.vertex_to_attribute_map);
      std::vector<u32> opp(conn.ct.num_corners());
      for (u32 c = 0; c < conn.ct.num_corners(); ++c) opp[c] = t.opposite(c, conn.ct);
      blob_put(dump, tag + ".opp", opp);
    }
  }
  return "";
}

}  // namespace orc
