// oracle/orc_decode.cpp — TEST INFRASTRUCTURE (see oracle.hpp header).
// The attribute section read BACKWARDS: what a decoder that has finished the connectivity stage (corner tables, seam tables,
// corners_of_edgebreaker) recovers from the bytes encode_attributes wrote — entropy decoding, inverse prediction transforms and the
// prediction schemes re-run on ALREADY DECODED values only, then dequantization.
//
// Source of the inverse: the reference's own decoder for this layer is not built (`// mod attribute;` is commented out of
// decode/mod.rs:5-6, decode/attribute/* is a prototype of an older layout, inverse_prediction_transform/oct_orthogonal.rs:40 is
// `unimplemented!()`).  What the reference does hold is used as it stands — decode/entropy/rans.rs:58-69,106-127 and
// decode/entropy/symbol_coding.rs:125-210 (orc_entropy.cpp) — and everything above the entropy layer is the inverse of the ENCODER
// spec, function by function (cited below), the way the Draco bitstream this encoder targets is decoded:
//   wrapped difference  orig = clamp(pred) + corr, wrapped back into [min, max]          (wrapped_difference.rs:54-99 inverted)
//   difference          orig = pred + corr                                                (difference.rs:26-34 inverted)
//   oct-orthogonal      centre, diamond inversion and quarter turns decided by the PREDICTION alone, as in the map; undo them in
//                       reverse order                                                      (oct_orthogonal.rs:23-74 inverted)
//   predictions         the encoder's functions with "contains(v)" = "v has been decoded" and the choices the encoder recorded
//                       (normal flips, texture-coordinate orientations) read from their rABS streams instead of being derived
//   dequantization      v = min + q · (range / (2^bits - 1)), octahedral (u, v) → unit vector    (the Draco dequantizers)
// A prediction that would need a value the decoder does not have yet is reported as an error: that is the property the round trip
// pins (the encoder only ever looked at vertices coded before the current one).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>

#include "oracle.hpp"

namespace orc {

namespace {

inline i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
inline i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
inline i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
inline i64 wadd64(i64 a, i64 b) { return (i64)((u64)a + (u64)b); }
inline i64 wsub64(i64 a, i64 b) { return (i64)((u64)a - (u64)b); }
inline i64 wmul64(i64 a, i64 b) { return (i64)((u64)a * (u64)b); }
inline i64 wabs64(i64 a) { return a < 0 ? (i64)(0 - (u64)a) : a; }
inline i64 wdiv64(i64 a, i64 b) { if (a == std::numeric_limits<i64>::min() && b == -1) return a; return a / b; }
inline i32 from_positive(u32 s) { return (s & 1u) ? (i32)(0u - ((s >> 1) + 1u)) : (i32)(s >> 1); }   // utils/mod.rs:152-158 inverted
inline i32 f32_to_i32_sat(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return std::numeric_limits<i32>::max();
  if (f <= -2147483648.0f) return std::numeric_limits<i32>::min();
  return (i32)f;
}

struct Reader {
  const u8* p; size_t n, at = 0;
  bool ok = true;
  u8 r8() { if (at >= n) { ok = false; return 0; } return p[at++]; }
  u32 r32() { u32 v = 0; for (int k = 0; k < 4; ++k) v |= (u32)r8() << (8 * k); return v; }
  float rf32() { u32 b = r32(); float f; std::memcpy(&f, &b, 4); return f; }
  u64 leb() { u64 v = 0; u32 sh = 0; u8 b; do { b = r8(); v |= (u64)(b & 0x7F) << sh; sh += 7; } while ((b & 0x80) && ok && sh < 70); return v; }
};

// geom.rs:40-91 + :137-157 as orc_attribute.cpp has them (the normal predictor quantizes its own prediction)
inline void octahedral_transform_f32(float x, float y, float z, float& ou, float& ov) {
  float abs_sum = std::fabs(x) + std::fabs(y) + std::fabs(z);
  float u = y / abs_sum, v = z / abs_sum;
  if (x < 0.0f) {
    float u_out = (u < 0.0f) ? std::fabs(v) - 1.0f : 1.0f - std::fabs(v);
    float v_out = (v < 0.0f) ? std::fabs(u) - 1.0f : 1.0f - std::fabs(u);
    u = u_out; v = v_out;
  }
  ou = u; ov = v;
}
inline void into_faithful(i32 u, i32 v, i32& ox, i32& oy) {
  const i32 max = 255, half = max / 2;
  i32 x = u, y = v;
  if ((u == 0 && v == 0) || (u == 255 && v == 0) || (u == 0 && v == 255)) { ox = 255; oy = 255; return; }
  else if (u == 0 && v > 127) y = half - (v - half);
  else if (u == max && v < half) y = half + (half - v);
  else if (v == max && u < half) x = half + (half - u);
  else if (v == 0 && u > half) x = half - (u - half);
  ox = x; oy = y;
}
inline void oct_quantize_f32(float x, float y, float z, i32& qx, i32& qy) {
  float u, v;
  octahedral_transform_f32(x, y, z, u, v);
  into_faithful(f32_to_i32_sat((u + 1.0f) * 127.0f), f32_to_i32_sat((v + 1.0f) * 127.0f), qx, qy);
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

// the diamond inversion of oct_orthogonal.rs:35-45 applied to one point (centred coordinates)
inline void invert_diamond(i32& a, i32& b) {
  const i32 one = 127;
  auto sgn = [](i32 v) -> i32 { return v > 0 ? 1 : (v < 0 ? -1 : 0); };
  const i32 a0 = a;
  const i32 q = -sgn(wmul(a, b));
  a = wadd(wmul(q, b), wmul(sgn(a0), one));
  b = wadd(wmul(q, a0), wmul(sgn(b), one));
}

// The inverse of that inversion for a point with two non-zero coordinates.  The reference's formula multiplies by sign() and so is not
// its own inverse on the axes (sign(0) = 0): points of the square's boundary are sent ONTO an axis and cannot be brought back by the
// same formula.  The involution the format defines (Draco's InvertDiamond: reflect across the diamond edge of the point's quadrant,
// quadrant chosen with >= / <=) agrees with the reference's forward map wherever both coordinates are non-zero and inverts it there.
inline void invert_diamond_involution(i32& s, i32& t) {
  const i32 one = 127;
  i32 sign_s, sign_t;
  if (s >= 0 && t >= 0) { sign_s = 1; sign_t = 1; }
  else if (s <= 0 && t <= 0) { sign_s = -1; sign_t = -1; }
  else { sign_s = s > 0 ? 1 : -1; sign_t = t > 0 ? 1 : -1; }
  const i32 corner_s = sign_s * one, corner_t = sign_t * one;
  i32 us = t + t - corner_t, ut = s + s - corner_s;
  if (sign_s * sign_t >= 0) { us = -us; ut = -ut; }
  s = (us + corner_s) / 2;
  t = (ut + corner_t) / 2;
}

}  // namespace

// oct_orthogonal.rs:23-74 inverted: (prediction, correction) → original octahedral coordinates
void oct_orthogonal_inverse(const i32* pred_in, const i32* corr, i32* orig) {
  const i32 one = 127;
  auto iabs = [](i32 v) -> i32 { return v < 0 ? (i32)(0u - (u32)v) : v; };
  i32 p0 = wsub(pred_in[0], one), p1 = wsub(pred_in[1], one);
  const bool inverted = wadd(iabs(p0), iabs(p1)) > one;
  if (inverted) invert_diamond(p0, p1);
  int turns = 0;
  if (!(p0 == 0 && p1 == 0)) while (p0 >= 0 || p1 > 0) { i32 t = p0; p0 = (i32)(0u - (u32)p1); p1 = t; ++turns; }
  // corr = orig' - pred' (+255 when negative), orig' ∈ [-127, 127]
  i32 o0 = wadd(corr[0], p0), o1 = wadd(corr[1], p1);
  if (o0 > one) o0 = wsub(o0, 255);
  if (o1 > one) o1 = wsub(o1, 255);
  for (int k = 0; k < (4 - turns % 4) % 4; ++k) { i32 t = o0; o0 = (i32)(0u - (u32)o1); o1 = t; }   // the remaining quarter turns of a full circle
  if (inverted) invert_diamond_involution(o0, o1);   // (the prediction above went through the ENCODER's formula: the correction is relative to it)
  orig[0] = wadd(o0, one); orig[1] = wadd(o1, one);
}

std::string decode_attributes(const u8* data, size_t len, const ConnOutput& conn, std::vector<DecodedAttribute>& out, size_t* consumed) {
  Reader r{data, len};
  const u32 n_atts = r.r8();                                         // encode/attribute/mod.rs:26
  struct Dec { u8 att_dec_id, domain, traversal; };
  std::vector<Dec> decs(n_atts);
  for (u32 i = 0; i < n_atts; ++i) { decs[i].att_dec_id = r.r8(); decs[i].domain = r.r8(); decs[i].traversal = r.r8(); }   // :30-39
  out.assign(n_atts, DecodedAttribute{});
  for (u32 i = 0; i < n_atts; ++i) {                                 // :43-57
    DecodedAttribute& a = out[i];
    if (r.r8() != 1) return "attribute decoder with more than one attribute";
    a.type = (AttType)r.r8(); a.ctype = (CompType)r.r8(); a.ncomp = r.r8();
    if (r.r8() != 0) return "normalized flag set";
    a.id = r.r8();
    a.port = r.r8();
    a.domain = decs[i].domain;
  }
  if (!r.ok) return "NotEnoughData (attribute headers)";
  for (u32 i = 0; i < n_atts; ++i) {
    DecodedAttribute& a = out[i];
    a.scheme = r.r8(); a.transform = r.r8();                         // attribute_encoder.rs:159-160
    if (r.r8() != 1) return "rans_encoding flag not set";             // :344
    TableView tv;
    tv.ct = &conn.ct;
    tv.at = (i > 0 && i - 1 < conn.att_tables.size()) ? &conn.att_tables[i - 1] : nullptr;   // all_inclusive_corner_table.rs:31-45
    a.seq = compute_sequence(tv, conn.corners_of_edgebreaker, false);
    const size_t n = a.seq.size();
    const int N = a.port == 3 ? 2 : a.ncomp;                         // octahedral quantization portabilizes 3 → 2 components
    a.ncomp_port = N;
    std::vector<u32> symbols;
    size_t used = 0;
    std::string e = decode_symbols_direct(data + r.at, len - r.at, n * N, symbols, &used);
    if (!e.empty()) return "attribute " + std::to_string(i) + ": " + e;
    r.at += used;
    // scheme-dependent metadata order (attribute_encoder.rs:362-386)
    i32 t_min = 0, t_max = 0;
    auto read_transform_info = [&] {
      if (a.transform == 1) { t_min = (i32)r.r32(); t_max = (i32)r.r32(); }          // wrapped_difference.rs:95-98
      else if (a.transform == 3) { if (r.r32() != 255 || r.r32() != 127) e = "unexpected oct-orthogonal metadata"; }
    };
    std::vector<u8> bits;
    auto read_rabs = [&](size_t count) -> std::string {
      const u8 zp = r.r8();
      const u64 nbytes = r.leb();
      if (!r.ok || r.at + nbytes > len) return "NotEnoughData (rABS block)";
      std::string er = rabs_decode_stream(data + r.at, (size_t)nbytes, zp, count, bits);
      r.at += (size_t)nbytes;
      // the encoder pushed these bits FIRST TO LAST (mesh_normal_prediction.rs:154-157, …texture_coordinates.rs:257-259) and an ANS
      // decoder pops the last one first: bit k of the encoder's sequence is the (count-1-k)-th one decoded
      std::reverse(bits.begin(), bits.end());
      return er;
    };
    if (a.scheme == 6) {                                             // MeshNormalPrediction: transform info, zero_prob, flips
      read_transform_info();
      e = read_rabs(n);
      if (!e.empty()) return "attribute " + std::to_string(i) + " flips: " + e;
    } else if (a.scheme == 5) {                                      // texture coordinates: count, zero_prob, orientation transitions, transform info
      const u32 count = r.r32();
      e = read_rabs(count);
      if (!e.empty()) return "attribute " + std::to_string(i) + " orientations: " + e;
      // the stream codes "same as the next one" looking backwards from `true` (mesh_prediction_for_texture_coordinates.rs:241-256):
      // rebuild the orientations
      std::vector<u8> orient(count);
      bool last = true;
      for (size_t k = count; k-- > 0;) { if (!bits[k]) last = !last; orient[k] = last ? 1 : 0; }
      bits.swap(orient);
      read_transform_info();
    } else {
      read_transform_info();
    }
    if (!e.empty()) return e;
    // portabilization metadata (:384-386)
    float q_min[4] = {0, 0, 0, 0}, q_range = 0;
    int q_bits = 0;
    if (a.port == 2) { for (int k = 0; k < a.ncomp; ++k) q_min[k] = r.rf32(); q_range = r.rf32(); q_bits = r.r8(); }   // quantization_coordinate_wise.rs:56-59
    else if (a.port == 3) { if (r.r8() != 8) return "octahedral quantization bits != 8"; }                            // octahedral_quantization.rs:43
    if (!r.ok) return "NotEnoughData (attribute " + std::to_string(i) + " metadata)";

    // ---- inverse prediction + inverse transform, entry by entry ------------------------------------------------
    const DecodedAttribute* pos = nullptr;
    if (a.scheme == 5 || a.scheme == 6) {
      for (u32 j = 0; j < i; ++j) if (out[j].type == Position) { pos = &out[j]; break; }
      if (!pos || pos->ncomp_port != 3) return "attribute " + std::to_string(i) + " needs a decoded 3-component Position attribute";
    }
    // values by attribute-table vertex (all corners of a vertex carry the same value); decoded[v] = its sequence index + 1
    std::vector<i32> val((size_t)tv.num_vertices() * N, 0);
    std::vector<u32> when(tv.num_vertices(), 0);
    auto have = [&](u32 v, size_t k) { return when[v] != 0 && when[v] - 1 < k; };
    auto pos_of = [&](u32 corner, i64* o) {                            // get_position_for_vertex: the parent's value at this corner's point
      const u32 v = conn.ct.vertex_idx(corner);
      const i32* q = pos->by_vertex.data() + (size_t)v * 3;
      o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
    };
    i32 max_diff = 0;
    if (a.transform == 1) max_diff = wadd(1, wsub(t_max, t_min));
    size_t next_bit = 0;
    a.portable.resize(n * N);
    for (size_t k = 0; k < n; ++k) {
      const u32 c = a.seq[k];
      i32 pred[4] = {0, 0, 0, 0};
      auto previous_value = [&] { if (k > 0) { const i32* v = val.data() + (size_t)tv.vertex_idx(a.seq[k - 1]) * N; for (int j = 0; j < N; ++j) pred[j] = v[j]; } };
      if (a.scheme == 1) {                                           // mesh_parallelogram_prediction.rs:186-237
        const u32 opp = tv.opposite(c);
        bool done = false;
        if (opp != NONE) {
          const u32 ov = tv.vertex_idx(opp), nv = tv.vertex_idx(TableView::next(c)), pv = tv.vertex_idx(TableView::previous(c));
          if (have(ov, k) && have(nv, k) && have(pv, k)) {
            for (int j = 0; j < N; ++j) pred[j] = wsub(wadd(val[(size_t)nv * N + j], val[(size_t)pv * N + j]), val[(size_t)ov * N + j]);
            done = true;
          }
        }
        if (!done) previous_value();
      } else if (a.scheme == 0) {                                    // delta_prediction.rs:56-71
        previous_value();
      } else if (a.scheme == 6) {                                    // mesh_normal_prediction.rs:22-44,75-144
        i64 pc[3];
        pos_of(c, pc);
        auto face_normal = [&](u32 cc, i64* acc) {
          i64 pn[3], pp[3];
          pos_of(TableView::next(cc), pn); pos_of(TableView::previous(cc), pp);
          i32 dn[3], dp[3];
          for (int j = 0; j < 3; ++j) { dn[j] = wsub((i32)pn[j], (i32)pc[j]); dp[j] = wsub((i32)pp[j], (i32)pc[j]); }
          acc[0] = wadd64(acc[0], wsub(wmul(dn[1], dp[2]), wmul(dn[2], dp[1])));
          acc[1] = wadd64(acc[1], wsub(wmul(dn[2], dp[0]), wmul(dn[0], dp[2])));
          acc[2] = wadd64(acc[2], wsub(wmul(dn[0], dp[1]), wmul(dn[1], dp[0])));
        };
        u32 curr = c;
        for (;;) { u32 l = tv.swing_left(curr); if (l == NONE) break; curr = l; if (curr == c) break; }
        const u32 start = curr;
        i64 sum[3] = {0, 0, 0};
        face_normal(curr, sum);
        for (;;) { u32 rr = tv.swing_right(curr); if (rr == NONE) break; curr = rr; if (curr == start) break; face_normal(curr, sum); }
        const i64 upper = 1ll << 29;
        const i64 abs_sum = wadd64(wadd64(wabs64(sum[0]), wabs64(sum[1])), wabs64(sum[2]));
        if (abs_sum > upper) { const i64 q = abs_sum / upper; for (int j = 0; j < 3; ++j) sum[j] = wdiv64(sum[j], q); }
        const i32 o3[3] = {(i32)sum[0], (i32)sum[1], (i32)sum[2]};
        i32 p0 = 0, p1 = 0;
        if (!(o3[0] == 0 && o3[1] == 0 && o3[2] == 0)) oct_quantize_f32((float)o3[0], (float)o3[1], (float)o3[2], p0, p1);
        if (bits[k]) { p0 = wmul(p0, -1); p1 = wmul(p1, -1); }        // the encoder's flip choice (Q8), read back
        pred[0] = p0; pred[1] = p1;
      } else if (a.scheme == 5) {                                    // mesh_prediction_for_texture_coordinates.rs:51-81,107-219
        const u32 nc = TableView::next(c), pc = TableView::previous(c);
        const u32 nv = tv.vertex_idx(nc), pv = tv.vertex_idx(pc);
        auto fallback = [&] { if (have(nv, k)) { pred[0] = val[(size_t)nv * 2]; pred[1] = val[(size_t)nv * 2 + 1]; } else previous_value(); };
        bool done = false;
        if (have(nv, k) && have(pv, k)) {
          const i64 next_uv[2] = {val[(size_t)nv * 2], val[(size_t)nv * 2 + 1]}, prev_uv[2] = {val[(size_t)pv * 2], val[(size_t)pv * 2 + 1]};
          if (next_uv[0] == prev_uv[0] && next_uv[1] == prev_uv[1]) { pred[0] = (i32)prev_uv[0]; pred[1] = (i32)prev_uv[1]; done = true; }
          else {
            i64 cp[3], np[3], pp[3];
            pos_of(c, cp); pos_of(nc, np); pos_of(pc, pp);
            i64 pn[3];
            for (int j = 0; j < 3; ++j) pn[j] = wsub64(pp[j], np[j]);
            const u64 pn2 = (u64)wadd64(wadd64(wmul64(pn[0], pn[0]), wmul64(pn[1], pn[1])), wmul64(pn[2], pn[2]));
            if (pn2 != 0) {
              i64 cn[3];
              for (int j = 0; j < 3; ++j) cn[j] = wsub64(cp[j], np[j]);
              const i64 cdp = wadd64(wadd64(wmul64(pn[0], cn[0]), wmul64(pn[1], cn[1])), wmul64(pn[2], cn[2]));
              const i64 pn_uv[2] = {wsub64(prev_uv[0], next_uv[0]), wsub64(prev_uv[1], next_uv[1])};
              const i64 I64MAX = std::numeric_limits<i64>::max();
              const i64 n_uv_absmax = std::max(wabs64(next_uv[0]), wabs64(next_uv[1]));
              const i64 pn_uv_absmax = std::max(wabs64(pn_uv[0]), wabs64(pn_uv[1]));
              const i64 pn_absmax = std::max(std::max(wabs64(pn[0]), wabs64(pn[1])), wabs64(pn[2]));
              if (!(n_uv_absmax > wdiv64(I64MAX, (i64)pn2)) && !(wabs64(cdp) > wdiv64(I64MAX, pn_uv_absmax)) && !(wabs64(cdp) > wdiv64(I64MAX, pn_absmax))) {
                i64 x_uv[2], x_pos[3], cxv[3];
                for (int j = 0; j < 2; ++j) x_uv[j] = wadd64(wmul64(next_uv[j], (i64)pn2), wmul64(pn_uv[j], cdp));
                for (int j = 0; j < 3; ++j) x_pos[j] = wadd64(np[j], wdiv64(wmul64(pn[j], cdp), (i64)pn2));
                for (int j = 0; j < 3; ++j) cxv[j] = wsub64(cp[j], x_pos[j]);
                const u64 cx2 = (u64)wadd64(wadd64(wmul64(cxv[0], cxv[0]), wmul64(cxv[1], cxv[1])), wmul64(cxv[2], cxv[2]));
                const u64 norm = int_sqrt(cx2 * pn2);
                const i64 cx_uv[2] = {wmul64(pn_uv[1], (i64)norm), wmul64((i64)(0 - (u64)pn_uv[0]), (i64)norm)};
                if (next_bit >= bits.size()) return "attribute " + std::to_string(i) + ": orientation stream exhausted";
                const bool first = bits[next_bit++] != 0;              // the encoder's choice (dist0 < dist1), read back
                for (int j = 0; j < 2; ++j) pred[j] = (i32)wdiv64(first ? wadd64(x_uv[j], cx_uv[j]) : wsub64(x_uv[j], cx_uv[j]), (i64)pn2);
                done = true;
              }
            }
          }
        }
        if (!done) fallback();
      } else {
        return "unknown prediction scheme " + std::to_string(a.scheme);
      }
      // ---- inverse prediction transform ----
      i32 orig[4];
      if (a.transform == 1) {                                        // wrapped_difference.rs:54-99 inverted
        for (int j = 0; j < N; ++j) {
          i32 p = pred[j];
          if (t_min <= t_max) p = p < t_min ? t_min : (p > t_max ? t_max : p);
          i32 v = wadd(p, from_positive(symbols[k * N + j]));
          if (v > t_max) v = wsub(v, max_diff); else if (v < t_min) v = wadd(v, max_diff);
          orig[j] = v;
        }
      } else if (a.transform == 0) {                                 // difference.rs:26-34 inverted
        for (int j = 0; j < N; ++j) orig[j] = wadd(pred[j], from_positive(symbols[k * N + j]));
      } else if (a.transform == 3) {
        const i32 corr[2] = {(i32)symbols[k * 2], (i32)symbols[k * 2 + 1]};
        oct_orthogonal_inverse(pred, corr, orig);
      } else {
        return "unknown prediction transform " + std::to_string(a.transform);
      }
      const u32 v = tv.vertex_idx(c);
      for (int j = 0; j < N; ++j) { val[(size_t)v * N + j] = orig[j]; a.portable[k * N + j] = orig[j]; }
      when[v] = (u32)k + 1;
    }
    if (a.scheme == 5 && next_bit != bits.size()) return "attribute " + std::to_string(i) + ": " + std::to_string(bits.size() - next_bit) + " orientation bits left over";
    // positions are looked up by UNIVERSAL vertex by the attributes that follow
    if (a.type == Position) {
      a.by_vertex.assign((size_t)conn.ct.num_vertices() * 3, 0);
      if (N == 3) for (size_t k = 0; k < n; ++k) { const u32 v = conn.ct.vertex_idx(a.seq[k]); for (int j = 0; j < 3; ++j) a.by_vertex[(size_t)v * 3 + j] = a.portable[k * 3 + j]; }
    }
    // ---- dequantization ----
    a.values.assign(n * a.ncomp, 0.0f);
    if (a.port == 2) {
      const float maxq = (float)(u64)((1ull << q_bits) - 1ull);
      const float delta = q_range / maxq;
      for (size_t k = 0; k < n; ++k) for (int j = 0; j < a.ncomp; ++j) a.values[k * a.ncomp + j] = q_min[j] + (float)a.portable[k * N + j] * delta;
    } else if (a.port == 3) {
      for (size_t k = 0; k < n; ++k) {
        // octahedral (u, v) ∈ [0, 254]² → unit vector: the inverse of geom.rs:40-91 on the quantization grid of octahedral_quantization.rs:49-64
        float u = (float)a.portable[k * 2] / 127.0f - 1.0f, v = (float)a.portable[k * 2 + 1] / 127.0f - 1.0f;
        float x = 1.0f - std::fabs(u) - std::fabs(v);
        float y = u, z = v;
        if (x < 0.0f) { const float ya = y, za = z; y = (ya < 0.0f ? -1.0f : 1.0f) * (1.0f - std::fabs(za)); z = (za < 0.0f ? -1.0f : 1.0f) * (1.0f - std::fabs(ya)); }
        const float nrm = std::sqrt(x * x + y * y + z * z);
        if (nrm > 0.0f) { x /= nrm; y /= nrm; z /= nrm; }
        a.values[k * 3] = x; a.values[k * 3 + 1] = y; a.values[k * 3 + 2] = z;
      }
    } else {                                                         // ToBits: the portable values are the values
      for (size_t k = 0; k < n * (size_t)a.ncomp; ++k) std::memcpy(&a.values[k], &a.portable[k], 4);
    }
  }
  if (consumed) *consumed = r.at;
  return "";
}

}  // namespace orc
