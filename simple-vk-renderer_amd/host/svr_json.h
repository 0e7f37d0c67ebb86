// svr_json.h — a small JSON reader for the glTF loader (svr_gltf.cpp).
//
// The reference parses glTF with fastgltf (simdjson underneath, src/vk_loader.cpp:171-191); neither is
// in this tree, and the loader needs only: objects, arrays, strings (with escapes), numbers, booleans,
// null.  Numbers are kept as double (strtod: correctly rounded), which round-trips every float32 that
// a writer prints with 9 significant digits.
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace svrjson {

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;  // insertion order kept

  bool is_null() const { return kind == Null; }
  bool is_object() const { return kind == Object; }
  bool is_array() const { return kind == Array; }
  bool is_number() const { return kind == Number; }
  bool is_string() const { return kind == String; }
  size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
  // object member or a shared null value
  const Value& operator[](const char* key) const {
    static const Value none;
    if (kind != Object) return none;
    for (const auto& kv : obj)
      if (kv.first == key) return kv.second;
    return none;
  }
  const Value& operator[](size_t i) const {
    static const Value none;
    return (kind == Array && i < arr.size()) ? arr[i] : none;
  }
  bool has(const char* key) const { return !(*this)[key].is_null(); }
  double number_or(double d) const { return kind == Number ? num : d; }
  long long int_or(long long d) const { return kind == Number ? (long long)num : d; }
  std::string string_or(const std::string& d) const { return kind == String ? str : d; }
};

class Parser {
 public:
  Parser(const char* begin, const char* end) : p_(begin), end_(end) {}
  bool parse(Value& out, std::string* err) {
    skip();
    if (!value(out, 0)) {
      if (err) *err = err_ + " at byte " + std::to_string((long long)(p_ - begin()));
      return false;
    }
    skip();
    if (p_ != end_) {
      if (err) *err = "trailing characters after the JSON document";
      return false;
    }
    return true;
  }

 private:
  const char* p_;
  const char* end_;
  const char* begin_ = nullptr;
  std::string err_;
  const char* begin() { return begin_ ? begin_ : p_; }

  bool fail(const char* m) {
    err_ = m;
    return false;
  }
  void skip() {
    if (!begin_) begin_ = p_;
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) p_++;
  }
  bool literal(const char* s) {
    size_t n = std::strlen(s);
    if ((size_t)(end_ - p_) < n || std::memcmp(p_, s, n) != 0) return fail("bad literal");
    p_ += n;
    return true;
  }
  static void utf8(std::string& s, unsigned cp) {
    if (cp < 0x80) {
      s += (char)cp;
    } else if (cp < 0x800) {
      s += (char)(0xC0 | (cp >> 6));
      s += (char)(0x80 | (cp & 0x3F));
    } else if (cp < 0x10000) {
      s += (char)(0xE0 | (cp >> 12));
      s += (char)(0x80 | ((cp >> 6) & 0x3F));
      s += (char)(0x80 | (cp & 0x3F));
    } else {
      s += (char)(0xF0 | (cp >> 18));
      s += (char)(0x80 | ((cp >> 12) & 0x3F));
      s += (char)(0x80 | ((cp >> 6) & 0x3F));
      s += (char)(0x80 | (cp & 0x3F));
    }
  }
  bool hex4(unsigned& v) {
    if (end_ - p_ < 4) return fail("short \\u escape");
    v = 0;
    for (int i = 0; i < 4; i++) {
      char c = *p_++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
      else return fail("bad \\u escape");
    }
    return true;
  }
  bool string(std::string& s) {
    p_++;  // opening quote
    while (p_ < end_) {
      char c = *p_++;
      if (c == '"') return true;
      if (c != '\\') {
        s += c;
        continue;
      }
      if (p_ >= end_) break;
      char e = *p_++;
      switch (e) {
        case '"': s += '"'; break;
        case '\\': s += '\\'; break;
        case '/': s += '/'; break;
        case 'b': s += '\b'; break;
        case 'f': s += '\f'; break;
        case 'n': s += '\n'; break;
        case 'r': s += '\r'; break;
        case 't': s += '\t'; break;
        case 'u': {
          unsigned cp;
          if (!hex4(cp)) return false;
          if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
            p_ += 2;
            unsigned lo;
            if (!hex4(lo)) return false;
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          utf8(s, cp);
          break;
        }
        default: return fail("bad escape");
      }
    }
    return fail("unterminated string");
  }
  bool value(Value& v, int depth) {
    if (depth > 256) return fail("nesting too deep");
    if (p_ >= end_) return fail("unexpected end");
    char c = *p_;
    if (c == '{') {
      v.kind = Value::Object;
      p_++;
      skip();
      if (p_ < end_ && *p_ == '}') {
        p_++;
        return true;
      }
      for (;;) {
        skip();
        if (p_ >= end_ || *p_ != '"') return fail("expected a member name");
        std::string key;
        if (!string(key)) return false;
        skip();
        if (p_ >= end_ || *p_ != ':') return fail("expected ':'");
        p_++;
        skip();
        v.obj.emplace_back(std::move(key), Value());
        if (!value(v.obj.back().second, depth + 1)) return false;
        skip();
        if (p_ < end_ && *p_ == ',') {
          p_++;
          continue;
        }
        if (p_ < end_ && *p_ == '}') {
          p_++;
          return true;
        }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      v.kind = Value::Array;
      p_++;
      skip();
      if (p_ < end_ && *p_ == ']') {
        p_++;
        return true;
      }
      for (;;) {
        skip();
        v.arr.emplace_back();
        if (!value(v.arr.back(), depth + 1)) return false;
        skip();
        if (p_ < end_ && *p_ == ',') {
          p_++;
          continue;
        }
        if (p_ < end_ && *p_ == ']') {
          p_++;
          return true;
        }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') {
      v.kind = Value::String;
      return string(v.str);
    }
    if (c == 't') {
      v.kind = Value::Bool;
      v.b = true;
      return literal("true");
    }
    if (c == 'f') {
      v.kind = Value::Bool;
      v.b = false;
      return literal("false");
    }
    if (c == 'n') {
      v.kind = Value::Null;
      return literal("null");
    }
    if (c == '-' || (c >= '0' && c <= '9')) {
      const char* q = p_;
      while (q < end_ && (*q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E' || (*q >= '0' && *q <= '9'))) q++;
      std::string tmp(p_, q);  // strtod needs a terminator
      char* stop = nullptr;
      v.num = std::strtod(tmp.c_str(), &stop);
      if (stop == tmp.c_str() || *stop != 0) return fail("bad number");
      v.kind = Value::Number;
      p_ = q;
      return true;
    }
    return fail("unexpected character");
  }
};

inline bool parse(const std::string& text, Value& out, std::string* err) {
  Parser ps(text.data(), text.data() + text.size());
  return ps.parse(out, err);
}

}  // namespace svrjson
