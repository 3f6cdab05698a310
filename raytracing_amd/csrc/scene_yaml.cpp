// scene_yaml.cpp -- see scene_yaml.hpp.
#include "scene_yaml.hpp"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rt {
namespace yaml {

const Node* Node::find(const std::string& key) const
{
    if (kind != Map) return nullptr;
    for (const auto& f : fields)
        if (f.first == key) return &f.second;
    return nullptr;
}

namespace {

struct Line {
    int indent;
    std::string text;  // without indentation, comment and trailing blanks
    int number;        // 1-based, for messages
};

struct Parser {
    std::vector<Line> lines;
    size_t pos = 0;
    std::string error;

    // Nesting of flow sequences and of blocks: the schema needs 3 levels; a hostile file of a few
    // hundred thousand '[' or '- - - ...' must end in an error, not in a stack overflow.
    static constexpr int kMaxDepth = 64;
    int depth = 0;
    struct Nest {
        int& d;
        explicit Nest(int& depth) : d(depth) { d++; }
        ~Nest() { d--; }
    };

    bool fail(int line, const std::string& what)
    {
        if (error.empty()) error = "line " + std::to_string(line) + ": " + what;
        return false;
    }

    static bool is_dash(const std::string& t)
    {
        return !t.empty() && t[0] == '-' && (t.size() == 1 || t[1] == ' ' || t[1] == '\t');
    }

    // "key:" or "key: value" with a plain key; returns the key length or 0.
    static size_t key_length(const std::string& t)
    {
        if (t.empty() || t[0] == '[' || t[0] == '-' || t[0] == '"' || t[0] == '\'') return 0;
        size_t i = 0;
        while (i < t.size() && (isalnum((unsigned char)t[i]) || t[i] == '_' || t[i] == '-' || t[i] == '.')) i++;
        if (i == 0 || i >= t.size() || t[i] != ':') return 0;
        if (i + 1 < t.size() && t[i + 1] != ' ' && t[i + 1] != '\t') return 0;
        return i;
    }

    static std::string trim(const std::string& s)
    {
        size_t a = 0, b = s.size();
        while (a < b && (s[a] == ' ' || s[a] == '\t')) a++;
        while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) b--;
        return s.substr(a, b - a);
    }

    bool parse_flow(const std::string& t, size_t& i, int line, Node* out)
    {
        Nest nest(depth);
        if (depth > kMaxDepth) return fail(line, "nesting too deep");
        while (i < t.size() && (t[i] == ' ' || t[i] == '\t')) i++;
        if (i < t.size() && t[i] == '[') {
            out->kind = Node::Sequence;
            i++;
            for (;;) {
                while (i < t.size() && (t[i] == ' ' || t[i] == '\t')) i++;
                if (i >= t.size()) return fail(line, "unterminated flow sequence");
                if (t[i] == ']') { i++; return true; }
                Node item;
                if (!parse_flow(t, i, line, &item)) return false;
                out->items.push_back(std::move(item));
                while (i < t.size() && (t[i] == ' ' || t[i] == '\t')) i++;
                if (i < t.size() && t[i] == ',') { i++; continue; }
                if (i < t.size() && t[i] == ']') { i++; return true; }
                return fail(line, "expected ',' or ']' in flow sequence");
            }
        }
        size_t start = i;
        while (i < t.size() && t[i] != ',' && t[i] != ']') i++;
        out->kind = Node::Scalar;
        out->scalar = trim(t.substr(start, i - start));
        if (out->scalar.empty()) return fail(line, "empty entry in flow sequence");
        return true;
    }

    bool parse_inline(const std::string& t, int line, Node* out)
    {
        if (t[0] == '[') {
            size_t i = 0;
            if (!parse_flow(t, i, line, out)) return false;
            if (!trim(t.substr(i)).empty()) return fail(line, "text after flow sequence");
            return true;
        }
        out->kind = Node::Scalar;
        out->scalar = t;
        if (t.size() >= 2 && (t[0] == '"' || t[0] == '\'') && t.back() == t[0])
            out->scalar = t.substr(1, t.size() - 2);
        return true;
    }

    bool parse_block(Node* out)
    {
        Nest nest(depth);
        if (depth > kMaxDepth) return fail(lines[pos].number, "nesting too deep");
        const Line first = lines[pos];
        if (is_dash(first.text)) return parse_sequence(first.indent, out);
        if (key_length(first.text)) return parse_map(first.indent, out);
        pos++;
        return parse_inline(first.text, first.number, out);
    }

    bool parse_sequence(int indent, Node* out)
    {
        out->kind = Node::Sequence;
        while (pos < lines.size() && lines[pos].indent == indent && is_dash(lines[pos].text)) {
            Line& l = lines[pos];
            size_t k = 1;
            while (k < l.text.size() && (l.text[k] == ' ' || l.text[k] == '\t')) k++;
            Node item;
            if (k >= l.text.size()) {  // "-" alone: the item is the more-indented block below
                pos++;
                if (pos < lines.size() && lines[pos].indent > indent) {
                    if (!parse_block(&item)) return false;
                }
            } else {  // "- rest": rest starts a node whose column is indent + k
                l.indent = indent + (int)k;
                l.text = l.text.substr(k);
                if (!parse_block(&item)) return false;
            }
            out->items.push_back(std::move(item));
        }
        if (pos < lines.size() && lines[pos].indent > indent)
            return fail(lines[pos].number, "unexpected indentation");
        return true;
    }

    bool parse_map(int indent, Node* out)
    {
        out->kind = Node::Map;
        while (pos < lines.size() && lines[pos].indent == indent) {
            const Line l = lines[pos];
            const size_t kl = key_length(l.text);
            if (!kl) return fail(l.number, "expected 'key:'");
            const std::string key = l.text.substr(0, kl);
            const std::string rest = trim(l.text.substr(kl + 1));
            Node value;
            pos++;
            if (!rest.empty()) {
                if (!parse_inline(rest, l.number, &value)) return false;
            } else if (pos < lines.size() &&
                       (lines[pos].indent > indent ||
                        (lines[pos].indent == indent && is_dash(lines[pos].text)))) {
                if (!parse_block(&value)) return false;
            }
            out->fields.emplace_back(key, std::move(value));
        }
        if (pos < lines.size() && lines[pos].indent > indent)
            return fail(lines[pos].number, "unexpected indentation");
        return true;
    }
};

}  // namespace

bool parse(const std::string& text, Node* out, std::string* error)
{
    Parser p;
    int number = 0;
    size_t at = 0;
    while (at <= text.size()) {
        size_t end = text.find('\n', at);
        if (end == std::string::npos) end = text.size();
        std::string raw = text.substr(at, end - at);
        at = end + 1;
        number++;
        // strip a comment: '#' at the start or after a blank
        for (size_t i = 0; i < raw.size(); i++)
            if (raw[i] == '#' && (i == 0 || raw[i - 1] == ' ' || raw[i - 1] == '\t')) {
                raw.resize(i);
                break;
            }
        int indent = 0;
        while ((size_t)indent < raw.size() && raw[indent] == ' ') indent++;
        if ((size_t)indent < raw.size() && raw[indent] == '\t') {
            if (error) *error = "line " + std::to_string(number) + ": tab used for indentation";
            return false;
        }
        std::string body = Parser::trim(raw);
        if (body.empty() || body == "---" || body == "...") continue;
        p.lines.push_back({indent, body, number});
    }
    *out = Node();
    if (p.lines.empty()) return true;
    bool ok = p.parse_block(out);
    if (ok && p.pos < p.lines.size()) ok = p.fail(p.lines[p.pos].number, "unexpected content");
    if (!ok && error) *error = p.error;
    return ok;
}

bool to_float(const Node& n, float* out)
{
    if (n.kind != Node::Scalar || n.scalar.empty()) return false;
    const std::string& s = n.scalar;
    if (s == ".inf" || s == ".Inf" || s == ".INF" || s == "+.inf") { *out = INFINITY; return true; }
    if (s == "-.inf" || s == "-.Inf" || s == "-.INF") { *out = -INFINITY; return true; }
    if (s == ".nan" || s == ".NaN" || s == ".NAN") { *out = NAN; return true; }
    if (isspace((unsigned char)s[0])) return false;
    char* end = nullptr;
    const float v = strtof(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') return false;
    *out = v;
    return true;
}

bool to_int(const Node& n, int* out)
{
    if (n.kind != Node::Scalar || n.scalar.empty()) return false;
    const std::string& s = n.scalar;
    if (isspace((unsigned char)s[0])) return false;
    char* end = nullptr;
    const long v = strtol(s.c_str(), &end, 10);
    if (end == s.c_str() || *end != '\0') return false;
    if (v < -2147483647L - 1 || v > 2147483647L) return false;
    *out = (int)v;
    return true;
}

std::string format_float(float v)
{
    if (std::isnan(v)) return ".nan";
    if (std::isinf(v)) return v < 0 ? "-.inf" : ".inf";
    if (v == 0.0f) return std::signbit(v) ? "-0" : "0";
    // shortest digit string d.ddd x 10^e that reads back to v
    char buf[48];
    for (int prec = 0; prec <= 8; prec++) {
        snprintf(buf, sizeof buf, "%.*e", prec, (double)v);
        if (strtof(buf, nullptr) == v) break;
    }
    std::string digits;
    const char* p = buf;
    const bool neg = *p == '-';
    if (neg) p++;
    for (; *p && *p != 'e'; p++)
        if (*p != '.') digits += *p;
    const int e10 = atoi(p + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const int nd = (int)digits.size();
    std::string out = neg ? "-" : "";
    if (e10 >= -5 && e10 < 9) {  // positional notation: 10 -> "10", 0.5672846f -> "0.5672846"
        if (e10 >= nd - 1) {
            out += digits + std::string((size_t)(e10 - (nd - 1)), '0');
        } else if (e10 >= 0) {
            out += digits.substr(0, (size_t)e10 + 1) + "." + digits.substr((size_t)e10 + 1);
        } else {
            out += "0." + std::string((size_t)(-e10 - 1), '0') + digits;
        }
    } else {  // scientific, at least two exponent digits like an ostream
        out += digits.substr(0, 1);
        if (nd > 1) out += "." + digits.substr(1);
        char ebuf[16];
        snprintf(ebuf, sizeof ebuf, "e%c%02d", e10 < 0 ? '-' : '+', e10 < 0 ? -e10 : e10);
        out += ebuf;
    }
    return out;
}

}  // namespace yaml
}  // namespace rt
