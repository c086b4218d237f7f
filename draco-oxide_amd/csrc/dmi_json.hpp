// dmi_json.hpp — the JSON a glTF transcode needs: a document tree that survives a round trip (object members keep their order, number tokens keep
// their text — nothing is re-formatted through a double), edits with the semantics of an insertion-ordered map (assign keeps a key's place, a new
// key goes to the end), a compact writer ("," and ":" without spaces).  Host only.  The reference goes through serde_json (io/gltf/decode.rs,
// io/gltf/encode.rs:362-400); byte equality of the JSON chunk with the reference is not part of the bit-exact contract — the
// embedded .drc blobs are.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace dmi {
namespace json {

struct Value {
  enum Kind : uint8_t { Null, False, True, Number, String, Array, Object };
  Kind kind = Null;
  std::string text;                                   // Number: the token as written; String: the decoded text (UTF-8)
  std::vector<Value> items;                           // Array
  std::vector<std::pair<std::string, Value>> members; // Object, document order

  static Value number(uint64_t v) { Value x; x.kind = Number; x.text = std::to_string(v); return x; }
  static Value string(std::string s) { Value x; x.kind = String; x.text = std::move(s); return x; }
  static Value array() { Value x; x.kind = Array; return x; }
  static Value object() { Value x; x.kind = Object; return x; }

  bool is_object() const { return kind == Object; }
  bool is_array() const { return kind == Array; }
  const Value* find(const char* key) const {
    if (kind != Object) return nullptr;
    for (const auto& m : members) if (m.first == key) return &m.second;
    return nullptr;
  }
  Value* find(const char* key) { return const_cast<Value*>(static_cast<const Value*>(this)->find(key)); }
  // dict[key] = v: an existing key keeps its place
  Value& set(const std::string& key, Value v) {
    for (auto& m : members) if (m.first == key) { m.second = std::move(v); return m.second; }
    kind = Object;
    members.emplace_back(key, std::move(v));
    return members.back().second;
  }
  void erase(const char* key) {
    for (size_t i = 0; i < members.size(); ++i) if (members[i].first == key) { members.erase(members.begin() + (long)i); return; }
  }
  // a non-negative integer token ("12"; no fraction, no exponent: what glTF asks of an index, a count, an offset)
  bool as_index(uint64_t* out) const {
    if (kind != Number || text.empty() || text.size() > 19) return false;
    uint64_t v = 0;
    for (char c : text) { if (c < '0' || c > '9') return false; v = v * 10 + (uint64_t)(c - '0'); }
    *out = v;
    return true;
  }
};

namespace detail {
struct Parser {
  const char* p;
  const char* end;
  std::string err;
  int depth = 0;
  bool fail(const char* what) { if (err.empty()) err = what; return false; }
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  static void put_utf8(std::string& s, uint32_t c) {
    if (c < 0x80) s.push_back((char)c);
    else if (c < 0x800) { s.push_back((char)(0xC0 | (c >> 6))); s.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) { s.push_back((char)(0xE0 | (c >> 12))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F))); }
    else { s.push_back((char)(0xF0 | (c >> 18))); s.push_back((char)(0x80 | ((c >> 12) & 0x3F))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F))); }
  }
  bool hex4(uint32_t* out) {
    if (end - p < 4) return fail("truncated \\u escape");
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) {
      const char c = *p++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
      else return fail("bad \\u escape");
    }
    *out = v;
    return true;
  }
  bool string(std::string& out) {   // *p == '"'
    ++p;
    for (;;) {
      const char* q = p;
      while (q < end && *q != '"' && *q != '\\' && (unsigned char)*q >= 0x20) ++q;
      out.append(p, q);
      p = q;
      if (p >= end) return fail("unterminated string");
      if (*p == '"') { ++p; return true; }
      if ((unsigned char)*p < 0x20) return fail("control character in a string");
      ++p;   // backslash
      if (p >= end) return fail("unterminated string");
      const char c = *p++;
      switch (c) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          uint32_t u;
          if (!hex4(&u)) return false;
          if (u >= 0xD800 && u < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
            const char* save = p;
            p += 2;
            uint32_t lo;
            if (!hex4(&lo)) return false;
            if (lo >= 0xDC00 && lo < 0xE000) u = 0x10000 + ((u - 0xD800) << 10) + (lo - 0xDC00);
            else p = save;
          }
          put_utf8(out, u);
          break;
        }
        default: return fail("bad escape");
      }
    }
  }
  bool number(Value& v) {
    const char* q = p;
    if (q < end && *q == '-') ++q;
    if (q >= end || *q < '0' || *q > '9') return fail("bad number");
    if (*q == '0') ++q; else while (q < end && *q >= '0' && *q <= '9') ++q;
    if (q < end && *q == '.') { ++q; if (q >= end || *q < '0' || *q > '9') return fail("bad number"); while (q < end && *q >= '0' && *q <= '9') ++q; }
    if (q < end && (*q == 'e' || *q == 'E')) { ++q; if (q < end && (*q == '+' || *q == '-')) ++q; if (q >= end || *q < '0' || *q > '9') return fail("bad number"); while (q < end && *q >= '0' && *q <= '9') ++q; }
    v.kind = Value::Number;
    v.text.assign(p, q);
    p = q;
    return true;
  }
  bool value(Value& v) {
    ws();
    if (p >= end) return fail("unexpected end of the document");
    if (++depth > 200) return fail("nesting too deep");
    bool ok = true;
    switch (*p) {
      case '{': {
        ++p;
        v.kind = Value::Object;
        ws();
        if (p < end && *p == '}') { ++p; break; }
        for (;;) {
          ws();
          if (p >= end || *p != '"') { ok = fail("expected a member name"); break; }
          std::string key;
          if (!string(key)) { ok = false; break; }
          ws();
          if (p >= end || *p != ':') { ok = fail("expected ':'"); break; }
          ++p;
          v.members.emplace_back(std::move(key), Value());
          if (!value(v.members.back().second)) { ok = false; break; }
          ws();
          if (p < end && *p == ',') { ++p; continue; }
          if (p < end && *p == '}') { ++p; break; }
          ok = fail("expected ',' or '}'");
          break;
        }
        break;
      }
      case '[': {
        ++p;
        v.kind = Value::Array;
        ws();
        if (p < end && *p == ']') { ++p; break; }
        for (;;) {
          v.items.emplace_back();
          if (!value(v.items.back())) { ok = false; break; }
          ws();
          if (p < end && *p == ',') { ++p; continue; }
          if (p < end && *p == ']') { ++p; break; }
          ok = fail("expected ',' or ']'");
          break;
        }
        break;
      }
      case '"': v.kind = Value::String; ok = string(v.text); break;
      case 't': if (end - p >= 4 && !std::memcmp(p, "true", 4)) { v.kind = Value::True; p += 4; } else ok = fail("bad literal"); break;
      case 'f': if (end - p >= 5 && !std::memcmp(p, "false", 5)) { v.kind = Value::False; p += 5; } else ok = fail("bad literal"); break;
      case 'n': if (end - p >= 4 && !std::memcmp(p, "null", 4)) { v.kind = Value::Null; p += 4; } else ok = fail("bad literal"); break;
      default: ok = number(v); break;
    }
    --depth;
    return ok;
  }
};
inline void write_string(const std::string& s, std::string& out) {
  out.push_back('"');
  for (unsigned char c : s) {
    switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\n': out += "\\n"; break;
      case '\r': out += "\\r"; break;
      case '\t': out += "\\t"; break;
      case '\b': out += "\\b"; break;
      case '\f': out += "\\f"; break;
      default:
        if (c < 0x20) { static const char* hex = "0123456789abcdef"; out += "\\u00"; out.push_back(hex[c >> 4]); out.push_back(hex[c & 15]); }
        else out.push_back((char)c);
    }
  }
  out.push_back('"');
}
}  // namespace detail

// the whole of [p, p + n) must be one JSON value (white space around it allowed)
inline bool parse(const char* p, size_t n, Value& out, std::string& err) {
  detail::Parser ps{p, p + n, {}, 0};
  if (!ps.value(out)) { err = ps.err; return false; }
  ps.ws();
  if (ps.p != ps.end) { err = "trailing characters after the document"; return false; }
  return true;
}
inline void write(const Value& v, std::string& out) {
  switch (v.kind) {
    case Value::Null: out += "null"; break;
    case Value::False: out += "false"; break;
    case Value::True: out += "true"; break;
    case Value::Number: out += v.text; break;
    case Value::String: detail::write_string(v.text, out); break;
    case Value::Array:
      out.push_back('[');
      for (size_t i = 0; i < v.items.size(); ++i) { if (i) out.push_back(','); write(v.items[i], out); }
      out.push_back(']');
      break;
    case Value::Object:
      out.push_back('{');
      for (size_t i = 0; i < v.members.size(); ++i) { if (i) out.push_back(','); detail::write_string(v.members[i].first, out); out.push_back(':'); write(v.members[i].second, out); }
      out.push_back('}');
      break;
  }
}

}  // namespace json
}  // namespace dmi
