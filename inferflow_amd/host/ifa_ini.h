// ifa_ini.h -- the reference's .ini dialect (sslib ConfigData as used by
// InferenceEngine::LoadConfig, src/transformer/inference_engine.cc:1412-1560):
//   [section] / key = value / ; # // comments / ${macro} expansion in values, with
//   ${config_dir} predefined as the directory of the file (trailing slash) and further
//   macros added by the caller (data_root_dir, model_name, global_model_dir).
// Section and key lookups are case-insensitive.
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace inferflow_amd {

class IniConfig {
public:
    bool Load(const std::string &path, std::string *err = nullptr)
    {
        std::ifstream f(path);
        if (!f) { if (err) *err = "cannot open " + path; return false; }
        std::stringstream ss; ss << f.rdbuf();
        size_t slash = path.find_last_of("/\\");
        AddMacro("config_dir", slash == std::string::npos ? std::string("./") : path.substr(0, slash + 1));
        return LoadText(ss.str(), err);
    }
    bool LoadText(const std::string &text, std::string *err = nullptr)
    {
        std::istringstream in(text);
        std::string line, section;
        int lineno = 0;
        while (std::getline(in, line)) {
            lineno++;
            std::string t = Trim(line);
            if (lineno == 1 && t.size() >= 3 && (unsigned char)t[0] == 0xEF) t = Trim(t.substr(3));   // UTF-8 BOM
            if (t.empty() || t[0] == ';' || t[0] == '#' || t.compare(0, 2, "//") == 0) continue;
            if (t[0] == '[') {
                size_t e = t.find(']');
                if (e == std::string::npos) { if (err) *err = "line " + std::to_string(lineno) + ": unterminated section"; return false; }
                section = Lower(Trim(t.substr(1, e - 1)));
                sections_.push_back(section);
                continue;
            }
            size_t eq = t.find('=');
            if (eq == std::string::npos) continue;       // tolerated, like the reference
            data_[section][Lower(Trim(t.substr(0, eq)))] = Trim(t.substr(eq + 1));
        }
        return true;
    }
    void AddMacro(const std::string &name, const std::string &value) { macros_[name] = value; }

    bool HasSection(const std::string &section) const { return data_.count(Lower(section)) != 0; }
    bool GetItem(const std::string &section, const std::string &key, std::string &out) const
    {
        auto s = data_.find(Lower(section));
        if (s == data_.end()) return false;
        auto k = s->second.find(Lower(key));
        if (k == s->second.end()) return false;
        out = Expand(k->second, 0);
        return true;
    }
    bool GetItem(const std::string &section, const std::string &key, int &out) const
    {
        std::string s; if (!GetItem(section, key, s) || s.empty()) return false;
        out = atoi(s.c_str()); return true;
    }
    bool GetItem(const std::string &section, const std::string &key, float &out) const
    {
        std::string s; if (!GetItem(section, key, s) || s.empty()) return false;
        out = (float)atof(s.c_str()); return true;
    }
    bool GetItem(const std::string &section, const std::string &key, bool &out) const
    {
        std::string s; if (!GetItem(section, key, s) || s.empty()) return false;
        s = Lower(s);
        out = s == "1" || s == "true" || s == "yes" || s == "on"; return true;
    }
    static std::string Trim(const std::string &s)
    {
        size_t a = 0, b = s.size();
        while (a < b && isspace((unsigned char)s[a])) a++;
        while (b > a && isspace((unsigned char)s[b - 1])) b--;
        return s.substr(a, b - a);
    }
    static std::string Lower(std::string s)
    {
        std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)tolower(c); });
        return s;
    }
    static std::vector<std::string> Split(const std::string &s, const std::string &seps)
    {
        std::vector<std::string> out; std::string cur;
        for (char c : s) {
            if (seps.find(c) != std::string::npos) { out.push_back(cur); cur.clear(); } else cur += c;
        }
        out.push_back(cur);
        return out;
    }

private:
    std::map<std::string, std::map<std::string, std::string>> data_;
    std::vector<std::string> sections_;
    std::map<std::string, std::string> macros_;

    std::string Expand(const std::string &v, int depth) const
    {
        if (depth > 8) return v;
        std::string out; size_t i = 0;
        while (i < v.size()) {
            if (v[i] == '$' && i + 1 < v.size() && v[i + 1] == '{') {
                size_t e = v.find('}', i + 2);
                if (e != std::string::npos) {
                    auto m = macros_.find(v.substr(i + 2, e - i - 2));
                    if (m != macros_.end()) { out += Expand(m->second, depth + 1); i = e + 1; continue; }
                }
            }
            out += v[i++];
        }
        return out;
    }
};

} // namespace inferflow_amd
