// A small TOML reader: the subset of TOML 1.0 that scene files use and a little more.
//
// Supported: comments; bare, quoted and dotted keys; [table] and [[array-of-tables]] headers at any depth (a
// sub-table header attaches to the LAST element of an enclosing array of tables); basic and literal strings (single
// and multi-line); integers (decimal with '_' separators, 0x/0o/0b); floats (fraction, exponent, inf, nan);
// booleans; arrays (nested, multi-line, trailing comma); inline tables.  Not supported: dates and times (reported as
// an error).  Duplicate keys and re-defined tables are errors, as in TOML.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rt {
namespace toml {

struct Value;
using Member = std::pair<std::string, std::unique_ptr<Value>>;

struct Value {
    enum Type { STRING, INTEGER, FLOAT, BOOLEAN, ARRAY, TABLE } type = TABLE;
    std::string s;
    int64_t i = 0;
    double f = 0.0;
    bool b = false;
    std::vector<std::unique_ptr<Value>> array;
    std::vector<Member> table;     // insertion order
    bool array_of_tables = false;  // ARRAY created by a [[header]]
    bool defined = false;          // TABLE named by a [header] or filled by an inline table / key = value
    bool inline_closed = false;    // inline table or static array: may not be extended later

    const Value *get(const std::string &key) const;
    Value *get(const std::string &key);
    bool is_number() const { return type == INTEGER || type == FLOAT; }
    double as_double() const { return type == INTEGER ? (double)i : f; }
    const char *type_name() const;
};

// Parses `text`; on failure returns nullptr and sets err to "line N: message".
std::unique_ptr<Value> parse(const std::string &text, std::string &err);

}  // namespace toml
}  // namespace rt
