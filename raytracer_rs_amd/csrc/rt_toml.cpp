// TOML-subset reader used by the scene loader (see rt_toml.hpp for what is covered).
#include "rt_toml.hpp"

#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace rt {
namespace toml {

const Value *Value::get(const std::string &key) const {
    for (const Member &m : table)
        if (m.first == key) return m.second.get();
    return nullptr;
}
Value *Value::get(const std::string &key) {
    for (Member &m : table)
        if (m.first == key) return m.second.get();
    return nullptr;
}
const char *Value::type_name() const {
    switch (type) {
        case STRING: return "string";
        case INTEGER: return "integer";
        case FLOAT: return "float";
        case BOOLEAN: return "boolean";
        case ARRAY: return "array";
        default: return "table";
    }
}

namespace {

struct Parser {
    const std::string &src;
    size_t pos = 0;
    int line = 1;
    std::string err;

    explicit Parser(const std::string &s) : src(s) {}

    bool fail(const std::string &msg) {
        if (err.empty()) err = "line " + std::to_string(line) + ": " + msg;
        return false;
    }
    bool eof() const { return pos >= src.size(); }
    char peek(size_t k = 0) const { return pos + k < src.size() ? src[pos + k] : '\0'; }

    void skip_ws() {
        while (!eof() && (peek() == ' ' || peek() == '\t')) pos++;
    }
    void skip_comment() {
        if (peek() == '#')
            while (!eof() && peek() != '\n') pos++;
    }
    // consumes a line ending (or end of input) after optional whitespace and comment
    bool expect_eol() {
        skip_ws();
        skip_comment();
        if (eof()) return true;
        if (peek() == '\r' && peek(1) == '\n') pos++;
        if (peek() == '\n') {
            pos++;
            line++;
            return true;
        }
        return fail("expected end of line");
    }
    // whitespace, newlines and comments (inside arrays)
    void skip_ws_nl() {
        for (;;) {
            skip_ws();
            skip_comment();
            if (peek() == '\r' && peek(1) == '\n') pos++;
            if (peek() == '\n') {
                pos++;
                line++;
                continue;
            }
            break;
        }
    }

    static bool bare_char(char c) {
        return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-';
    }

    bool parse_basic_string(std::string &out, bool multiline) {
        // opening quote(s) already consumed
        for (;;) {
            if (eof()) return fail("unterminated string");
            char c = src[pos++];
            if (c == '"') {
                if (!multiline) return true;
                if (peek() == '"' && peek(1) == '"') {
                    pos += 2;
                    // up to two extra quotes belong to the content
                    while (peek() == '"') { out.push_back('"'); pos++; }
                    return true;
                }
                out.push_back('"');
                continue;
            }
            if (c == '\n') {
                if (!multiline) return fail("newline in string");
                line++;
                out.push_back('\n');
                continue;
            }
            if (c == '\\') {
                if (eof()) return fail("unterminated escape");
                char e = src[pos++];
                switch (e) {
                    case 'b': out.push_back('\b'); break;
                    case 't': out.push_back('\t'); break;
                    case 'n': out.push_back('\n'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'r': out.push_back('\r'); break;
                    case '"': out.push_back('"'); break;
                    case '\\': out.push_back('\\'); break;
                    case 'u':
                    case 'U': {
                        int digits = e == 'u' ? 4 : 8;
                        uint32_t cp = 0;
                        for (int k = 0; k < digits; ++k) {
                            char h = peek();
                            int v = (h >= '0' && h <= '9') ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10
                                    : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : -1;
                            if (v < 0) return fail("bad unicode escape");
                            cp = cp * 16 + (uint32_t)v;
                            pos++;
                        }
                        if (cp < 0x80) out.push_back((char)cp);
                        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        break;
                    }
                    case '\n':
                    case ' ':
                    case '\t':
                    case '\r':
                        if (!multiline) return fail("bad escape");
                        // line-ending backslash: trim whitespace and newlines
                        if (e == '\n') line++;
                        while (!eof() && (peek() == ' ' || peek() == '\t' || peek() == '\n' || peek() == '\r')) {
                            if (peek() == '\n') line++;
                            pos++;
                        }
                        break;
                    default: return fail("bad escape");
                }
                continue;
            }
            out.push_back(c);
        }
    }

    bool parse_literal_string(std::string &out, bool multiline) {
        for (;;) {
            if (eof()) return fail("unterminated string");
            char c = src[pos++];
            if (c == '\'') {
                if (!multiline) return true;
                if (peek() == '\'' && peek(1) == '\'') {
                    pos += 2;
                    while (peek() == '\'') { out.push_back('\''); pos++; }
                    return true;
                }
                out.push_back('\'');
                continue;
            }
            if (c == '\n') {
                if (!multiline) return fail("newline in string");
                line++;
            }
            out.push_back(c);
        }
    }

    bool parse_string(std::string &out) {
        char q = peek();
        bool multi = peek(1) == q && peek(2) == q;
        pos += multi ? 3 : 1;
        if (multi) {  // a newline right after the opening delimiter is trimmed
            if (peek() == '\r' && peek(1) == '\n') pos++;
            if (peek() == '\n') { pos++; line++; }
        }
        return q == '"' ? parse_basic_string(out, multi) : parse_literal_string(out, multi);
    }

    bool parse_key_part(std::string &out) {
        skip_ws();
        if (peek() == '"' || peek() == '\'') {
            if (peek(1) == peek() && peek(2) == peek()) return fail("multi-line string used as key");
            return parse_string(out);
        }
        size_t start = pos;
        while (!eof() && bare_char(peek())) pos++;
        if (pos == start) return fail("expected a key");
        out.assign(src, start, pos - start);
        return true;
    }

    bool parse_dotted_key(std::vector<std::string> &parts) {
        for (;;) {
            std::string k;
            if (!parse_key_part(k)) return false;
            parts.push_back(std::move(k));
            skip_ws();
            if (peek() == '.') {
                pos++;
                continue;
            }
            return true;
        }
    }

    bool parse_number_or_date(Value &v) {
        size_t start = pos;
        while (!eof()) {
            char c = peek();
            if ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '+' || c == '-' ||
                c == '.' || c == '_' || c == ':')
                pos++;
            else
                break;
        }
        std::string tok(src, start, pos - start);
        if (tok.empty()) return fail("expected a value");
        if (tok == "true" || tok == "false") {
            v.type = Value::BOOLEAN;
            v.b = tok == "true";
            return true;
        }
        std::string body = tok;
        bool neg = false;
        if (body[0] == '+' || body[0] == '-') {
            neg = body[0] == '-';
            body.erase(0, 1);
        }
        if (body == "inf" || body == "nan") {
            v.type = Value::FLOAT;
            v.f = body == "inf" ? (neg ? -INFINITY : INFINITY) : NAN;
            return true;
        }
        if (body.empty()) return fail("malformed number '" + tok + "'");
        if (body.find(':') != std::string::npos ||
            (body.size() >= 10 && body[4] == '-' && body[7] == '-' && body.find('e') == std::string::npos))
            return fail("dates and times are not supported");
        // underscores must sit between digits
        std::string clean;
        for (size_t k = 0; k < body.size(); ++k) {
            if (body[k] == '_') {
                if (k == 0 || k + 1 >= body.size() || !isxdigit((unsigned char)body[k - 1]) || !isxdigit((unsigned char)body[k + 1]))
                    return fail("malformed number '" + tok + "'");
                continue;
            }
            clean.push_back(body[k]);
        }
        if (clean.size() > 2 && clean[0] == '0' && (clean[1] == 'x' || clean[1] == 'o' || clean[1] == 'b')) {
            if (tok[0] == '+' || tok[0] == '-') return fail("malformed number '" + tok + "'");
            int base = clean[1] == 'x' ? 16 : clean[1] == 'o' ? 8 : 2;
            errno = 0;
            char *end = nullptr;
            unsigned long long u = std::strtoull(clean.c_str() + 2, &end, base);
            if (errno || *end) return fail("malformed number '" + tok + "'");
            v.type = Value::INTEGER;
            v.i = (int64_t)u;
            return true;
        }
        bool is_float = clean.find_first_of(".eE") != std::string::npos;
        for (char c : clean)
            if (!((c >= '0' && c <= '9') || c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-'))
                return fail("malformed value '" + tok + "'");
        if (!(clean[0] >= '0' && clean[0] <= '9')) return fail("malformed number '" + tok + "'");
        if (is_float) {
            size_t dot = clean.find('.');
            if (dot != std::string::npos &&
                (dot + 1 >= clean.size() || !(clean[dot + 1] >= '0' && clean[dot + 1] <= '9')))
                return fail("malformed number '" + tok + "'");
            errno = 0;
            char *end = nullptr;
            double d = std::strtod(clean.c_str(), &end);
            if (*end) return fail("malformed number '" + tok + "'");
            v.type = Value::FLOAT;
            v.f = neg ? -d : d;
            return true;
        }
        if (clean.size() > 1 && clean[0] == '0') return fail("leading zero in integer '" + tok + "'");
        errno = 0;
        char *end = nullptr;
        const std::string digits = std::string(neg ? "-" : "") + clean;  // must outlive `end`, which points into it
        long long ll = std::strtoll(digits.c_str(), &end, 10);
        if (errno || *end) return fail("malformed number '" + tok + "'");
        v.type = Value::INTEGER;
        v.i = ll;
        return true;
    }

    // Arrays and inline tables nest; a file that nests them thousands deep is broken or hostile, and recursing into
    // it would overflow the stack instead of returning an error.
    static constexpr int MAX_VALUE_DEPTH = 64;
    int value_depth = 0;
    struct DepthGuard {
        int &d;
        explicit DepthGuard(int &depth) : d(depth) { ++d; }
        ~DepthGuard() { --d; }
    };

    bool parse_value(Value &v) {
        DepthGuard guard(value_depth);
        if (value_depth > MAX_VALUE_DEPTH) return fail("values are nested too deep");
        skip_ws();
        char c = peek();
        if (c == '"' || c == '\'') {
            v.type = Value::STRING;
            return parse_string(v.s);
        }
        if (c == '[') {
            pos++;
            v.type = Value::ARRAY;
            v.inline_closed = true;
            for (;;) {
                skip_ws_nl();
                if (eof()) return fail("unterminated array");
                if (peek() == ']') {
                    pos++;
                    return true;
                }
                auto elem = std::make_unique<Value>();
                if (!parse_value(*elem)) return false;
                v.array.push_back(std::move(elem));
                skip_ws_nl();
                if (peek() == ',') {
                    pos++;
                    continue;
                }
                if (peek() == ']') {
                    pos++;
                    return true;
                }
                return fail("expected ',' or ']' in array");
            }
        }
        if (c == '{') {
            pos++;
            v.type = Value::TABLE;
            v.defined = true;
            skip_ws();
            if (peek() == '}') {
                pos++;
                v.inline_closed = true;
                return true;
            }
            for (;;) {
                if (!parse_key_value(v)) return false;
                skip_ws();
                if (peek() == ',') {
                    pos++;
                    continue;
                }
                if (peek() == '}') {
                    pos++;
                    v.inline_closed = true;
                    return true;
                }
                return fail("expected ',' or '}' in inline table");
            }
        }
        return parse_number_or_date(v);
    }

    // key = value into `table` (dotted keys create intermediate tables)
    bool parse_key_value(Value &table) {
        std::vector<std::string> parts;
        if (!parse_dotted_key(parts)) return false;
        skip_ws();
        if (peek() != '=') return fail("expected '=' after key");
        pos++;
        Value *cur = &table;
        for (size_t k = 0; k + 1 < parts.size(); ++k) {
            Value *next = cur->get(parts[k]);
            if (!next) {
                auto t = std::make_unique<Value>();
                t->type = Value::TABLE;
                next = t.get();
                cur->table.emplace_back(parts[k], std::move(t));
            } else if (next->type != Value::TABLE || next->inline_closed) {
                return fail("key '" + parts[k] + "' is not a table");
            }
            cur = next;
        }
        if (cur->get(parts.back())) return fail("duplicate key '" + parts.back() + "'");
        auto val = std::make_unique<Value>();
        if (!parse_value(*val)) return false;
        cur->table.emplace_back(parts.back(), std::move(val));
        return true;
    }

    // resolves a [header] / [[header]] path; returns the table that following key/value lines fill
    Value *open_table(Value &root, const std::vector<std::string> &parts, bool is_array) {
        Value *cur = &root;
        for (size_t k = 0; k < parts.size(); ++k) {
            bool last = k + 1 == parts.size();
            Value *next = cur->get(parts[k]);
            if (!next) {
                auto nv = std::make_unique<Value>();
                if (last && is_array) {
                    nv->type = Value::ARRAY;
                    nv->array_of_tables = true;
                } else {
                    nv->type = Value::TABLE;
                }
                next = nv.get();
                cur->table.emplace_back(parts[k], std::move(nv));
            }
            if (last) {
                if (is_array) {
                    if (next->type != Value::ARRAY || !next->array_of_tables) {
                        fail("'" + parts[k] + "' is not an array of tables");
                        return nullptr;
                    }
                    auto elem = std::make_unique<Value>();
                    elem->type = Value::TABLE;
                    elem->defined = true;
                    Value *e = elem.get();
                    next->array.push_back(std::move(elem));
                    return e;
                }
                if (next->type != Value::TABLE || next->inline_closed) {
                    fail("'" + parts[k] + "' is not a table");
                    return nullptr;
                }
                if (next->defined) {
                    fail("table '" + parts[k] + "' defined twice");
                    return nullptr;
                }
                next->defined = true;
                return next;
            }
            if (next->type == Value::ARRAY) {
                if (!next->array_of_tables || next->array.empty()) {
                    fail("'" + parts[k] + "' is not an array of tables");
                    return nullptr;
                }
                next = next->array.back().get();
            } else if (next->type != Value::TABLE || next->inline_closed) {
                fail("'" + parts[k] + "' is not a table");
                return nullptr;
            }
            cur = next;
        }
        return cur;
    }

    bool parse_document(Value &root) {
        root.type = Value::TABLE;
        Value *current = &root;
        if (src.size() >= 3 && (unsigned char)src[0] == 0xEF && (unsigned char)src[1] == 0xBB && (unsigned char)src[2] == 0xBF) pos = 3;
        while (!eof()) {
            skip_ws();
            if (eof()) break;
            char c = peek();
            if (c == '#' || c == '\n' || c == '\r') {
                if (!expect_eol()) return false;
                continue;
            }
            if (c == '[') {
                bool is_array = peek(1) == '[';
                pos += is_array ? 2 : 1;
                std::vector<std::string> parts;
                if (!parse_dotted_key(parts)) return false;
                skip_ws();
                if (is_array) {
                    if (!(peek() == ']' && peek(1) == ']')) return fail("expected ']]'");
                    pos += 2;
                } else {
                    if (peek() != ']') return fail("expected ']'");
                    pos++;
                }
                current = open_table(root, parts, is_array);
                if (!current) return false;
                if (!expect_eol()) return false;
                continue;
            }
            if (!parse_key_value(*current)) return false;
            if (!expect_eol()) return false;
        }
        return true;
    }
};

}  // namespace

std::unique_ptr<Value> parse(const std::string &text, std::string &err) {
    Parser p(text);
    auto root = std::make_unique<Value>();
    if (!p.parse_document(*root)) {
        err = p.err.empty() ? "parse error" : p.err;
        return nullptr;
    }
    return root;
}

}  // namespace toml
}  // namespace rt
