// Minimal JSON helpers for the server shell: syntax validation of a request body and string
// escaping for the status document.  (The reference links nlohmann/json, an empty submodule in
// its checkout; the only JSON produced here are the fixed shapes of SURVEY §A.3.)
#pragma once
#include <cctype>
#include <cstdio>
#include <string>

namespace JsonMin {

class Validator {
    const std::string &s;
    size_t p = 0;
    int depth = 0;
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++; }
    bool lit(const char *w) {
        size_t n = 0;
        while (w[n]) n++;
        if (s.compare(p, n, w) != 0) return false;
        p += n;
        return true;
    }
    bool str() {
        if (p >= s.size() || s[p] != '"') return false;
        p++;
        while (p < s.size()) {
            unsigned char c = (unsigned char)s[p++];
            if (c == '"') return true;
            if (c < 0x20) return false;
            if (c == '\\') {
                if (p >= s.size()) return false;
                char e = s[p++];
                if (e == 'u') {
                    for (int i = 0; i < 4; i++)
                        if (p >= s.size() || !isxdigit((unsigned char)s[p++])) return false;
                } else if (!(e == '"' || e == '\\' || e == '/' || e == 'b' || e == 'f' || e == 'n' || e == 'r' || e == 't'))
                    return false;
            }
        }
        return false;
    }
    bool num() {
        size_t st = p;
        if (p < s.size() && s[p] == '-') p++;
        if (p >= s.size()) return false;
        if (s[p] == '0') p++;
        else if (isdigit((unsigned char)s[p])) while (p < s.size() && isdigit((unsigned char)s[p])) p++;
        else return false;
        if (p < s.size() && s[p] == '.') {
            p++;
            if (p >= s.size() || !isdigit((unsigned char)s[p])) return false;
            while (p < s.size() && isdigit((unsigned char)s[p])) p++;
        }
        if (p < s.size() && (s[p] == 'e' || s[p] == 'E')) {
            p++;
            if (p < s.size() && (s[p] == '+' || s[p] == '-')) p++;
            if (p >= s.size() || !isdigit((unsigned char)s[p])) return false;
            while (p < s.size() && isdigit((unsigned char)s[p])) p++;
        }
        return p > st;
    }
    bool value() {
        if (++depth > 512) return false;
        ws();
        bool ok = false;
        if (p >= s.size()) ok = false;
        else if (s[p] == '{') {
            p++;
            ws();
            if (p < s.size() && s[p] == '}') { p++; ok = true; }
            else
                for (;;) {
                    ws();
                    if (!str()) break;
                    ws();
                    if (p >= s.size() || s[p++] != ':') break;
                    if (!value()) break;
                    ws();
                    if (p < s.size() && s[p] == ',') { p++; continue; }
                    if (p < s.size() && s[p] == '}') { p++; ok = true; }
                    break;
                }
        } else if (s[p] == '[') {
            p++;
            ws();
            if (p < s.size() && s[p] == ']') { p++; ok = true; }
            else
                for (;;) {
                    if (!value()) break;
                    ws();
                    if (p < s.size() && s[p] == ',') { p++; continue; }
                    if (p < s.size() && s[p] == ']') { p++; ok = true; }
                    break;
                }
        } else if (s[p] == '"') ok = str();
        else if (s[p] == 't') ok = lit("true");
        else if (s[p] == 'f') ok = lit("false");
        else if (s[p] == 'n') ok = lit("null");
        else ok = num();
        depth--;
        return ok;
    }

public:
    explicit Validator(const std::string &text) : s(text) {}
    bool valid() {
        if (!value()) return false;
        ws();
        return p == s.size();
    }
};

inline bool isValid(const std::string &text) { return Validator(text).valid(); }

// "..." with the escapes nlohmann's dump() emits for these characters
inline std::string quote(const std::string &v) {
    std::string o = "\"";
    for (unsigned char c : v) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    snprintf(buf, sizeof buf, "\\u%04x", c);
                    o += buf;
                } else
                    o += (char)c;
        }
    }
    return o + "\"";
}

}   // namespace JsonMin
