// Minimal JSON reader/writer for the scene schema (host only).
// The reference parses scenes with nlohmann::json (gpu-version/parser.hpp:12-14,
// 505-506), which is not present on the target image; the scene schema only needs
// objects, arrays, numbers, strings and booleans, so this is a ~200-line recursive
// descent parser that reports line/column on error instead of throwing through
// the C ABI.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rtmi {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JsonValue> arr;
    // insertion-ordered object (scene files are small; linear lookup is fine)
    std::vector<std::pair<std::string, JsonValue>> obj;

    const JsonValue *find(const char *key) const {
        if (kind != Object) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_number() const { return kind == Number; }
    bool is_array() const { return kind == Array; }
    bool is_object() const { return kind == Object; }
    bool is_string() const { return kind == String; }
};

class JsonParser {
public:
    JsonParser(const char *text, size_t len) : p_(text), end_(text + len), begin_(text) {}

    // returns false and fills err on failure
    bool parse(JsonValue &out, std::string &err) {
        skip_ws();
        if (!value(out, 0)) {
            err = err_;
            return false;
        }
        skip_ws();
        if (p_ != end_) {
            fail("trailing characters after the top-level value");
            err = err_;
            return false;
        }
        return true;
    }

private:
    const char *p_, *end_, *begin_;
    std::string err_;
    static constexpr int kMaxDepth = 64;

    bool fail(const char *msg) {
        if (!err_.empty()) return false;
        int line = 1, col = 1;
        for (const char *q = begin_; q < p_ && q < end_; ++q) {
            if (*q == '\n') {
                ++line;
                col = 1;
            } else {
                ++col;
            }
        }
        char buf[256];
        snprintf(buf, sizeof buf, "JSON error at line %d column %d: %s", line, col, msg);
        err_ = buf;
        return false;
    }
    void skip_ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
    }
    bool literal(const char *lit) {
        size_t n = strlen(lit);
        if ((size_t)(end_ - p_) < n || memcmp(p_, lit, n) != 0) return fail("invalid literal");
        p_ += n;
        return true;
    }
    bool value(JsonValue &v, int depth) {
        if (depth > kMaxDepth) return fail("nesting too deep");
        if (p_ >= end_) return fail("unexpected end of input");
        switch (*p_) {
        case '{': return object(v, depth);
        case '[': return array(v, depth);
        case '"': v.kind = JsonValue::String; return string(v.str);
        case 't': v.kind = JsonValue::Bool; v.b = true; return literal("true");
        case 'f': v.kind = JsonValue::Bool; v.b = false; return literal("false");
        case 'n': v.kind = JsonValue::Null; return literal("null");
        default: return number(v);
        }
    }
    bool number(JsonValue &v) {
        const char *s = p_;
        if (p_ < end_ && *p_ == '-') ++p_;
        if (p_ >= end_ || !(*p_ >= '0' && *p_ <= '9')) return fail("invalid number");
        while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        if (p_ < end_ && *p_ == '.') {
            ++p_;
            if (p_ >= end_ || !(*p_ >= '0' && *p_ <= '9')) return fail("digit expected after '.'");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
            ++p_;
            if (p_ < end_ && (*p_ == '+' || *p_ == '-')) ++p_;
            if (p_ >= end_ || !(*p_ >= '0' && *p_ <= '9')) return fail("digit expected in exponent");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        std::string tmp(s, p_);
        v.kind = JsonValue::Number;
        v.num = strtod(tmp.c_str(), nullptr);
        return true;
    }
    bool string(std::string &out) {
        ++p_;  // opening quote
        out.clear();
        while (true) {
            if (p_ >= end_) return fail("unterminated string");
            unsigned char c = (unsigned char)*p_++;
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') {
                out.push_back((char)c);
                continue;
            }
            if (p_ >= end_) return fail("unterminated escape");
            char e = *p_++;
            switch (e) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                if (end_ - p_ < 4) return fail("short \\u escape");
                unsigned cp = 0;
                for (int i = 0; i < 4; ++i) {
                    char h = *p_++;
                    cp <<= 4;
                    if (h >= '0' && h <= '9') cp |= (unsigned)(h - '0');
                    else if (h >= 'a' && h <= 'f') cp |= (unsigned)(h - 'a' + 10);
                    else if (h >= 'A' && h <= 'F') cp |= (unsigned)(h - 'A' + 10);
                    else return fail("bad hex digit in \\u escape");
                }
                // UTF-8 encode (surrogate pairs are passed through as-is; file
                // names in scenes are ASCII)
                if (cp < 0x80) out.push_back((char)cp);
                else if (cp < 0x800) {
                    out.push_back((char)(0xC0 | (cp >> 6)));
                    out.push_back((char)(0x80 | (cp & 0x3F)));
                } else {
                    out.push_back((char)(0xE0 | (cp >> 12)));
                    out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                    out.push_back((char)(0x80 | (cp & 0x3F)));
                }
                break;
            }
            default: return fail("unknown escape");
            }
        }
    }
    bool array(JsonValue &v, int depth) {
        v.kind = JsonValue::Array;
        ++p_;
        skip_ws();
        if (p_ < end_ && *p_ == ']') {
            ++p_;
            return true;
        }
        while (true) {
            v.arr.emplace_back();
            skip_ws();
            if (!value(v.arr.back(), depth + 1)) return false;
            skip_ws();
            if (p_ >= end_) return fail("unterminated array");
            if (*p_ == ',') {
                ++p_;
                continue;
            }
            if (*p_ == ']') {
                ++p_;
                return true;
            }
            return fail("',' or ']' expected");
        }
    }
    bool object(JsonValue &v, int depth) {
        v.kind = JsonValue::Object;
        ++p_;
        skip_ws();
        if (p_ < end_ && *p_ == '}') {
            ++p_;
            return true;
        }
        while (true) {
            skip_ws();
            if (p_ >= end_ || *p_ != '"') return fail("object key expected");
            std::string key;
            if (!string(key)) return false;
            skip_ws();
            if (p_ >= end_ || *p_ != ':') return fail("':' expected");
            ++p_;
            skip_ws();
            v.obj.emplace_back(std::move(key), JsonValue());
            if (!value(v.obj.back().second, depth + 1)) return false;
            skip_ws();
            if (p_ >= end_) return fail("unterminated object");
            if (*p_ == ',') {
                ++p_;
                continue;
            }
            if (*p_ == '}') {
                ++p_;
                return true;
            }
            return fail("',' or '}' expected");
        }
    }
};

// shortest round-trip text of a float (so serialise -> parse reproduces the bits)
inline std::string json_float(float f) {
    char buf[64];
    for (int prec = 6; prec <= 9; ++prec) {
        snprintf(buf, sizeof buf, "%.*g", prec, (double)f);
        if (strtof(buf, nullptr) == f) break;
    }
    return buf;
}

inline std::string json_escape(const std::string &s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        if (c == '"') o += "\\\"";
        else if (c == '\\') o += "\\\\";
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c < 0x20) {
            char b[8];
            snprintf(b, sizeof b, "\\u%04x", c);
            o += b;
        } else o.push_back((char)c);
    }
    o += "\"";
    return o;
}

}  // namespace rtmi
