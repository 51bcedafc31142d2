// jellyfish_amd/include/jellyfish_amd/json_min.hpp
//
// Minimal JSON value (object / array / string / integer / double / bool / null)
// with a parser and a compact writer.  The reference vendors jsoncpp for its file
// header (include/jellyfish/json.h, lib/jsoncpp.cpp) and writes it with
// Json::FastWriter: no whitespace, object keys in sorted order
// (generic_file_header.hpp:88-111).  Readers only do keyed look-ups
// (file_header.hpp:35-108), so this small class is enough to be byte-compatible
// in style and fully compatible in content.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace jellyfish_amd {

class Json {
public:
  enum Type { Null, Bool, UInt, Int, Real, String, Array, Object };

  Json() : type_(Null) {}
  Json(bool b) : type_(Bool) { u_ = b; }
  Json(int v) : type_(v < 0 ? Int : UInt) { if(v < 0) i_ = v; else u_ = (uint64_t)v; }
  Json(unsigned v) : type_(UInt) { u_ = v; }
  Json(long v) : type_(v < 0 ? Int : UInt) { if(v < 0) i_ = v; else u_ = (uint64_t)v; }
  Json(unsigned long v) : type_(UInt) { u_ = v; }
  Json(unsigned long long v) : type_(UInt) { u_ = v; }
  Json(double d) : type_(Real) { d_ = d; }
  Json(const char* s) : type_(String), s_(s) {}
  Json(const std::string& s) : type_(String), s_(s) {}

  Type type() const { return type_; }
  bool is_null() const { return type_ == Null; }

  // object access (creates the object / member like jsoncpp's operator[])
  Json& operator[](const std::string& key) {
    if(type_ == Null) type_ = Object;
    if(type_ != Object) throw std::runtime_error("json: not an object");
    return obj_[key];
  }
  const Json& get(const std::string& key) const {
    static const Json null_value;
    if(type_ != Object) return null_value;
    auto it = obj_.find(key);
    return it == obj_.end() ? null_value : it->second;
  }
  bool has(const std::string& key) const { return type_ == Object && obj_.count(key); }
  const std::map<std::string, Json>& members() const { return obj_; }

  // array access
  void append(const Json& v) {
    if(type_ == Null) type_ = Array;
    if(type_ != Array) throw std::runtime_error("json: not an array");
    arr_.push_back(v);
  }
  size_t size() const { return type_ == Array ? arr_.size() : (type_ == Object ? obj_.size() : 0); }
  const Json& at(size_t i) const { static const Json n; return (type_ == Array && i < arr_.size()) ? arr_[i] : n; }
  void clear() { arr_.clear(); obj_.clear(); }
  void set_array() { type_ = Array; arr_.clear(); }

  uint64_t as_uint64(uint64_t dflt = 0) const {
    switch(type_) {
    case UInt: return u_;
    case Int: return (uint64_t)i_;
    case Real: return (uint64_t)d_;
    case Bool: return u_;
    default: return dflt;
    }
  }
  double as_double(double dflt = 0) const {
    switch(type_) {
    case UInt: return (double)u_;
    case Int: return (double)i_;
    case Real: return d_;
    default: return dflt;
    }
  }
  bool as_bool(bool dflt = false) const { return type_ == Bool ? (bool)u_ : (type_ == UInt ? u_ != 0 : dflt); }
  std::string as_string(const std::string& dflt = "") const { return type_ == String ? s_ : dflt; }

  // compact writer, FastWriter style
  std::string dump() const { std::string out; write(out); return out; }

  static Json parse(const char* begin, const char* end) {
    Parser p{begin, end};
    Json v = p.value();
    p.ws();
    if(p.p != p.e) throw std::runtime_error("json: trailing characters");
    return v;
  }

private:
  Type type_;
  union { uint64_t u_; int64_t i_; double d_; };
  std::string s_;
  std::vector<Json> arr_;
  std::map<std::string, Json> obj_;

  static void write_string(std::string& out, const std::string& s) {
    out += '"';
    for(unsigned char c : s) {
      switch(c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\b': out += "\\b"; break;
      case '\f': out += "\\f"; break;
      case '\n': out += "\\n"; break;
      case '\r': out += "\\r"; break;
      case '\t': out += "\\t"; break;
      default:
        if(c < 0x20) { char buf[8]; snprintf(buf, sizeof buf, "\\u%04X", c); out += buf; }
        else out += (char)c;
      }
    }
    out += '"';
  }
  void write(std::string& out) const {
    char buf[40];
    switch(type_) {
    case Null: out += "null"; break;
    case Bool: out += u_ ? "true" : "false"; break;
    case UInt: snprintf(buf, sizeof buf, "%llu", (unsigned long long)u_); out += buf; break;
    case Int: snprintf(buf, sizeof buf, "%lld", (long long)i_); out += buf; break;
    case Real: snprintf(buf, sizeof buf, "%.17g", d_); out += buf; break;
    case String: write_string(out, s_); break;
    case Array:
      out += '[';
      for(size_t i = 0; i < arr_.size(); ++i) { if(i) out += ','; arr_[i].write(out); }
      out += ']';
      break;
    case Object: {
      out += '{';
      bool first = true;
      for(const auto& kv : obj_) {
        if(!first) out += ',';
        first = false;
        write_string(out, kv.first);
        out += ':';
        kv.second.write(out);
      }
      out += '}';
      break;
    }
    }
  }

  struct Parser {
    const char* p; const char* e;
    void ws() { while(p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    [[noreturn]] void bad(const char* m) { throw std::runtime_error(std::string("json: ") + m); }
    Json value() {
      ws();
      if(p >= e) bad("unexpected end");
      switch(*p) {
      case '{': return object();
      case '[': return array();
      case '"': return Json(string());
      case 't': lit("true"); return Json(true);
      case 'f': lit("false"); return Json(false);
      case 'n': lit("null"); return Json();
      default: return number();
      }
    }
    void lit(const char* s) {
      size_t n = strlen(s);
      if((size_t)(e - p) < n || strncmp(p, s, n)) bad("bad literal");
      p += n;
    }
    Json number() {
      const char* s = p;
      bool neg = false, real = false;
      if(p < e && *p == '-') { neg = true; ++p; }
      while(p < e && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
        if(*p == '.' || *p == 'e' || *p == 'E') real = true;
        ++p;
      }
      if(p == s) bad("bad number");
      std::string t(s, p);
      if(real) return Json(strtod(t.c_str(), nullptr));
      if(neg) return Json((long)strtoll(t.c_str(), nullptr, 10));
      return Json((unsigned long long)strtoull(t.c_str(), nullptr, 10));
    }
    std::string string() {
      std::string out;
      ++p;
      while(p < e && *p != '"') {
        if(*p == '\\') {
          if(++p >= e) bad("bad escape");
          switch(*p) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
          case 'b': out += '\b'; break; case 'f': out += '\f'; break;
          case 'u': {
            if(e - p < 5) bad("bad \\u");
            unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            p += 4;
            if(cp < 0x80) out += (char)cp;
            else if(cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += *p;
          }
          ++p;
        } else out += *p++;
      }
      if(p >= e) bad("unterminated string");
      ++p;
      return out;
    }
    Json array() {
      Json a; a.set_array();
      ++p; ws();
      if(p < e && *p == ']') { ++p; return a; }
      while(true) {
        a.append(value());
        ws();
        if(p < e && *p == ',') { ++p; continue; }
        if(p < e && *p == ']') { ++p; return a; }
        bad("bad array");
      }
    }
    Json object() {
      Json o; o["_"]; o.clear();  // make it an (empty) object
      ++p; ws();
      if(p < e && *p == '}') { ++p; return o; }
      while(true) {
        ws();
        if(p >= e || *p != '"') bad("bad object key");
        std::string k = string();
        ws();
        if(p >= e || *p != ':') bad("missing ':'");
        ++p;
        o[k] = value();
        ws();
        if(p < e && *p == ',') { ++p; continue; }
        if(p < e && *p == '}') { ++p; return o; }
        bad("bad object");
      }
    }
  };
};

}  // namespace jellyfish_amd
