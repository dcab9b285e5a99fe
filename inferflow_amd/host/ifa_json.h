// ifa_json.h -- minimal JSON reader for model_spec.json / config.json / safetensors headers.
// Accepts // and /* */ comments and trailing commas (the reference's spec files are hand-written).
#pragma once
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

namespace inferflow_amd {

struct JsonValue {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;

    const JsonValue *Get(const std::string &key) const
    {
        if (type != Object) return nullptr;
        for (const auto &kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool GetString(const std::string &key, std::string &out) const
    {
        const JsonValue *v = Get(key);
        if (!v || v->type != String) return false;
        out = v->str; return true;
    }
    template <typename T> bool GetNumber(const std::string &key, T &out) const
    {
        const JsonValue *v = Get(key);
        if (!v) return false;
        if (v->type == Number) { out = (T)v->num; return true; }
        if (v->type == Bool) { out = (T)(v->b ? 1 : 0); return true; }
        return false;
    }
    bool GetBool(const std::string &key, bool &out) const
    {
        const JsonValue *v = Get(key);
        if (!v) return false;
        if (v->type == Bool) { out = v->b; return true; }
        if (v->type == Number) { out = v->num != 0; return true; }
        return false;
    }
};

class JsonParser {
public:
    bool Parse(const std::string &text, JsonValue &out, std::string *err = nullptr)
    {
        s_ = text.c_str(); n_ = text.size(); i_ = 0; err_.clear();
        bool ok = Value(out);
        if (ok) { Skip(); if (i_ != n_) { ok = false; err_ = "trailing characters"; } }
        if (!ok && err) *err = err_ + " at offset " + std::to_string(i_);
        return ok;
    }

private:
    const char *s_ = nullptr; size_t n_ = 0, i_ = 0; std::string err_;

    void Skip()
    {
        for (;;) {
            while (i_ < n_ && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) i_++;
            if (i_ + 1 < n_ && s_[i_] == '/' && s_[i_ + 1] == '/') { while (i_ < n_ && s_[i_] != '\n') i_++; continue; }
            if (i_ + 1 < n_ && s_[i_] == '/' && s_[i_ + 1] == '*') {
                i_ += 2;
                while (i_ + 1 < n_ && !(s_[i_] == '*' && s_[i_ + 1] == '/')) i_++;
                i_ = i_ + 2 <= n_ ? i_ + 2 : n_;
                continue;
            }
            break;
        }
    }
    bool Fail(const char *m) { if (err_.empty()) err_ = m; return false; }
    bool Value(JsonValue &v)
    {
        Skip();
        if (i_ >= n_) return Fail("unexpected end");
        char c = s_[i_];
        if (c == '{') return Obj(v);
        if (c == '[') return Arr(v);
        if (c == '"') { v.type = JsonValue::String; return Str(v.str); }
        if (n_ - i_ >= 4 && std::string(s_ + i_, 4) == "true") { v.type = JsonValue::Bool; v.b = true; i_ += 4; return true; }
        if (n_ - i_ >= 5 && std::string(s_ + i_, 5) == "false") { v.type = JsonValue::Bool; v.b = false; i_ += 5; return true; }
        if (n_ - i_ >= 4 && std::string(s_ + i_, 4) == "null") { v.type = JsonValue::Null; i_ += 4; return true; }
        char *end = nullptr;
        double d = strtod(s_ + i_, &end);
        if (end == s_ + i_) return Fail("unexpected character");
        v.type = JsonValue::Number; v.num = d; i_ = (size_t)(end - s_);
        return true;
    }
    static void Utf8(unsigned cp, std::string &out)
    {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    bool Str(std::string &out)
    {
        out.clear(); i_++;   // opening quote
        while (i_ < n_ && s_[i_] != '"') {
            char c = s_[i_++];
            if (c != '\\') { out += c; continue; }
            if (i_ >= n_) return Fail("bad escape");
            char e = s_[i_++];
            switch (e) {
            case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
            case 'b': out += '\b'; break; case 'f': out += '\f'; break;
            case 'u': {
                if (i_ + 4 > n_) return Fail("bad \\u escape");
                unsigned cp = (unsigned)strtoul(std::string(s_ + i_, 4).c_str(), nullptr, 16); i_ += 4;
                if (cp >= 0xD800 && cp < 0xDC00 && i_ + 6 <= n_ && s_[i_] == '\\' && s_[i_ + 1] == 'u') {
                    unsigned lo = (unsigned)strtoul(std::string(s_ + i_ + 2, 4).c_str(), nullptr, 16);
                    if (lo >= 0xDC00 && lo < 0xE000) { cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); i_ += 6; }
                }
                Utf8(cp, out); break;
            }
            default: out += e; break;   // \" \\ \/
            }
        }
        if (i_ >= n_) return Fail("unterminated string");
        i_++;
        return true;
    }
    bool Arr(JsonValue &v)
    {
        v.type = JsonValue::Array; i_++;
        for (;;) {
            Skip();
            if (i_ < n_ && s_[i_] == ']') { i_++; return true; }
            JsonValue e;
            if (!Value(e)) return false;
            v.arr.push_back(std::move(e));
            Skip();
            if (i_ < n_ && s_[i_] == ',') { i_++; continue; }
            if (i_ < n_ && s_[i_] == ']') { i_++; return true; }
            return Fail("expected , or ]");
        }
    }
    bool Obj(JsonValue &v)
    {
        v.type = JsonValue::Object; i_++;
        for (;;) {
            Skip();
            if (i_ < n_ && s_[i_] == '}') { i_++; return true; }
            if (i_ >= n_ || s_[i_] != '"') return Fail("expected a key");
            std::string key;
            if (!Str(key)) return false;
            Skip();
            if (i_ >= n_ || s_[i_] != ':') return Fail("expected :");
            i_++;
            JsonValue e;
            if (!Value(e)) return false;
            v.obj.emplace_back(std::move(key), std::move(e));
            Skip();
            if (i_ < n_ && s_[i_] == ',') { i_++; continue; }
            if (i_ < n_ && s_[i_] == '}') { i_++; return true; }
            return Fail("expected , or }");
        }
    }
};

} // namespace inferflow_amd
