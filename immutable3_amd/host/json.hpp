// json.hpp -- the small subset of JSON the reference's metadata files use (ujson in the reference):
// objects, arrays, strings, numbers, true/false/null.  `_table.meta` (core/.../Table.scala:27-43),
// `<col>_<id>.meta` (core/.../storage/Segment.scala:41-45).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace immutabledb {
namespace json {

struct Value {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj; // insertion-ordered, like ujson.Obj

    const Value &at(const std::string &key) const {
        for (const auto &kv : obj)
            if (kv.first == key) return kv.second;
        throw std::runtime_error("key not found: " + key); // NoSuchElementException in the reference
    }
    static Value string(const std::string &s) { Value v; v.kind = Str; v.str = s; return v; }
    static Value number(double d) { Value v; v.kind = Num; v.num = d; return v; }
    static Value array() { Value v; v.kind = Arr; return v; }
    static Value object() { Value v; v.kind = Obj; return v; }
    void put(const std::string &k, Value v) { obj.emplace_back(k, std::move(v)); }
};

class Parser {
  public:
    explicit Parser(const std::string &text) : s_(text) {}
    Value parse() {
        Value v = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    size_t i_ = 0;
    [[noreturn]] void fail(const char *msg) const { throw std::runtime_error(std::string("json: ") + msg + " at offset " + std::to_string(i_)); }
    void ws() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_; }
    Value value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        const char c = s_[i_];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') return Value::string(str());
        if (s_.compare(i_, 4, "true") == 0) { i_ += 4; Value v; v.kind = Value::Bool; v.b = true; return v; }
        if (s_.compare(i_, 5, "false") == 0) { i_ += 5; Value v; v.kind = Value::Bool; return v; }
        if (s_.compare(i_, 4, "null") == 0) { i_ += 4; return Value(); }
        char *end = nullptr;
        const double d = std::strtod(s_.c_str() + i_, &end);
        if (end == s_.c_str() + i_) fail("bad value");
        i_ = (size_t)(end - s_.c_str());
        return Value::number(d);
    }
    std::string str() {
        std::string out;
        ++i_;
        while (i_ < s_.size() && s_[i_] != '"') {
            if (s_[i_] == '\\' && i_ + 1 < s_.size()) {
                const char e = s_[++i_];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    const unsigned cp = (unsigned)std::strtoul(s_.substr(i_ + 1, 4).c_str(), nullptr, 16);
                    i_ += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += e;
                }
                ++i_;
            } else out += s_[i_++];
        }
        if (i_ >= s_.size()) fail("unterminated string");
        ++i_;
        return out;
    }
    Value array() {
        Value v = Value::array();
        ++i_;
        ws();
        if (i_ < s_.size() && s_[i_] == ']') { ++i_; return v; }
        for (;;) {
            v.arr.push_back(value());
            ws();
            if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
            if (i_ < s_.size() && s_[i_] == ']') { ++i_; return v; }
            fail("expected , or ]");
        }
    }
    Value object() {
        Value v = Value::object();
        ++i_;
        ws();
        if (i_ < s_.size() && s_[i_] == '}') { ++i_; return v; }
        for (;;) {
            ws();
            if (i_ >= s_.size() || s_[i_] != '"') fail("expected key");
            std::string k = str();
            ws();
            if (i_ >= s_.size() || s_[i_] != ':') fail("expected :");
            ++i_;
            v.obj.emplace_back(std::move(k), value());
            ws();
            if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
            if (i_ < s_.size() && s_[i_] == '}') { ++i_; return v; }
            fail("expected , or }");
        }
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }

inline void write(const Value &v, std::ostream &os) {
    switch (v.kind) {
    case Value::Null: os << "null"; break;
    case Value::Bool: os << (v.b ? "true" : "false"); break;
    case Value::Num:
        if (v.num == (double)(int64_t)v.num) os << (int64_t)v.num; // ujson renders integral doubles without ".0"
        else os << v.num;
        break;
    case Value::Str:
        os << '"';
        for (char c : v.str) {
            if (c == '"' || c == '\\') os << '\\' << c;
            else if (c == '\n') os << "\\n";
            else os << c;
        }
        os << '"';
        break;
    case Value::Arr:
        os << '[';
        for (size_t i = 0; i < v.arr.size(); ++i) { if (i) os << ','; write(v.arr[i], os); }
        os << ']';
        break;
    case Value::Obj:
        os << '{';
        for (size_t i = 0; i < v.obj.size(); ++i) {
            if (i) os << ',';
            os << '"' << v.obj[i].first << "\":";
            write(v.obj[i].second, os);
        }
        os << '}';
        break;
    }
}

inline std::string dump(const Value &v) { std::ostringstream os; write(v, os); return os.str(); }

} // namespace json
} // namespace immutabledb
