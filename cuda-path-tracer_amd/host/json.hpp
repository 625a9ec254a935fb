// json.hpp -- a small JSON reader (just enough for the scene grammar of assets/json_parser.cpp and for hip_pt's replay
// scripts; the reference uses nlohmann::json, which is not in this image).  Header-only.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace hip_pt {

struct Json;
using JsonPtr = std::shared_ptr<Json>;
struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<JsonPtr> arr;
  std::vector<std::pair<std::string, JsonPtr>> obj;

  [[nodiscard]] const Json* find(const std::string& key) const
  {
    for (const auto& kv : obj)
      if (kv.first == key) return kv.second.get();
    return nullptr;
  }
  [[nodiscard]] const Json& at(const std::string& key) const
  {
    const Json* j = find(key);
    if (!j) throw std::runtime_error("Json Parser: missing key " + key);
    return *j;
  }
  [[nodiscard]] float f() const
  {
    if (kind != Number) throw std::runtime_error("Json Parser: number expected");
    return (float)num;
  }
};

class JsonReader {
public:
  explicit JsonReader(std::string text) : s_(std::move(text)) {}
  JsonPtr parse()
  {
    JsonPtr v = value();
    ws();
    if (p_ != s_.size()) fail("trailing characters");
    return v;
  }

private:
  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("Json Parser: ") + what + " at offset " + std::to_string(p_)); }
  void ws()
  {
    while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) ++p_;
  }
  JsonPtr value()
  {
    ws();
    if (p_ >= s_.size()) fail("unexpected end");
    auto v = std::make_shared<Json>();
    const char c = s_[p_];
    if (c == '{') {
      v->kind = Json::Object;
      ++p_;
      ws();
      if (s_[p_] == '}') { ++p_; return v; }
      for (;;) {
        ws();
        const std::string key = string();
        ws();
        if (s_[p_++] != ':') fail("':' expected");
        v->obj.emplace_back(key, value());
        ws();
        if (s_[p_] == ',') { ++p_; continue; }
        if (s_[p_] == '}') { ++p_; break; }
        fail("',' or '}' expected");
      }
    } else if (c == '[') {
      v->kind = Json::Array;
      ++p_;
      ws();
      if (s_[p_] == ']') { ++p_; return v; }
      for (;;) {
        v->arr.push_back(value());
        ws();
        if (s_[p_] == ',') { ++p_; continue; }
        if (s_[p_] == ']') { ++p_; break; }
        fail("',' or ']' expected");
      }
    } else if (c == '"') {
      v->kind = Json::String;
      v->str = string();
    } else if (s_.compare(p_, 4, "true") == 0) {
      v->kind = Json::Bool; v->b = true; p_ += 4;
    } else if (s_.compare(p_, 5, "false") == 0) {
      v->kind = Json::Bool; p_ += 5;
    } else if (s_.compare(p_, 4, "null") == 0) {
      p_ += 4;
    } else {
      size_t used = 0;
      try { v->num = std::stod(s_.substr(p_), &used); } catch (...) { fail("value expected"); }
      v->kind = Json::Number;
      p_ += used;
    }
    return v;
  }
  std::string string()
  {
    if (s_[p_] != '"') fail("string expected");
    ++p_;
    std::string out;
    while (p_ < s_.size() && s_[p_] != '"') {
      if (s_[p_] == '\\' && p_ + 1 < s_.size()) {
        const char e = s_[p_ + 1];
        out += e == 'n' ? '\n' : e == 't' ? '\t' : e;
        p_ += 2;
      } else {
        out += s_[p_++];
      }
    }
    if (p_ >= s_.size()) fail("unterminated string");
    ++p_;
    return out;
  }
  std::string s_;
  size_t p_ = 0;
};

}  // namespace hip_pt
