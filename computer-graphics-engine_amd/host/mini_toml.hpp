// mini_toml.hpp — the subset of TOML that the reference's config files use
// (src/config.cpp:193-374 reads them with toml++): comments, bare / quoted keys, dotted
// table headers [a.b], arrays of tables [[cameras]], and values bool / integer / float /
// basic or literal string / (nested) array. No dates, no inline tables, no multi-line strings.
// A parse error is reported with its line number; nothing is guessed.
#pragma once
#include <cctype>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace mini_toml {

struct Value {
    enum Kind { NONE, BOOL, INT, FLOAT, STRING, ARRAY, TABLE } kind = NONE;
    bool b = false;
    long long i = 0;
    double f = 0.0;
    std::string s;
    std::vector<Value> a;
    std::map<std::string, Value> t;

    const Value* get(const std::string& key) const
    {
        if (kind != TABLE)
            return nullptr;
        auto it = t.find(key);
        return it == t.end() ? nullptr : &it->second;
    }
    // the reference's `as_boolean()->value_or(d)` / `value<T>().value_or(d)` idioms
    bool boolean_or(const std::string& key, bool d) const
    {
        const Value* v = get(key);
        return v && v->kind == BOOL ? v->b : d;
    }
    double number_or(const std::string& key, double d) const
    {
        const Value* v = get(key);
        return v && v->kind == FLOAT ? v->f : (v && v->kind == INT ? (double)v->i : d);
    }
    std::string string_or(const std::string& key, const std::string& d) const
    {
        const Value* v = get(key);
        return v && v->kind == STRING ? v->s : d;
    }
    bool is_number() const { return kind == INT || kind == FLOAT; }
    double number() const { return kind == FLOAT ? f : (double)i; }
};

class Parser {
public:
    explicit Parser(const std::string& text) : s_(text) {}
    // returns "" and fills `root`, or the error message
    std::string parse(Value& root)
    {
        root = Value();
        root.kind = Value::TABLE;
        Value* cur = &root;
        while (true) {
            skip_ws_and_comments(true);
            if (eof())
                return "";
            if (peek() == '[') {
                const bool arr = pos_ + 1 < s_.size() && s_[pos_ + 1] == '[';
                pos_ += arr ? 2 : 1;
                std::vector<std::string> path;
                std::string e = key_path(path);
                if (!e.empty())
                    return e;
                skip_ws();
                if (!(arr ? (consume(']') && consume(']')) : consume(']')))
                    return err("expected ] after a table name");
                cur = &root;
                for (size_t k = 0; k < path.size(); k++) {
                    Value& child = cur->t[path[k]];
                    const bool last = k + 1 == path.size();
                    if (last && arr) {
                        if (child.kind == Value::NONE)
                            child.kind = Value::ARRAY;
                        if (child.kind != Value::ARRAY)
                            return err("[[" + path[k] + "]] conflicts with an earlier key");
                        child.a.emplace_back();
                        child.a.back().kind = Value::TABLE;
                        cur = &child.a.back();
                    } else {
                        if (child.kind == Value::NONE)
                            child.kind = Value::TABLE;
                        if (child.kind == Value::ARRAY && !child.a.empty() && child.a.back().kind == Value::TABLE)
                            cur = &child.a.back(); // [cameras.sub] after [[cameras]]
                        else if (child.kind == Value::TABLE)
                            cur = &child;
                        else
                            return err("[" + path[k] + "] conflicts with an earlier key");
                    }
                }
                if (!end_of_line())
                    return err("unexpected characters after a table header");
                continue;
            }
            std::vector<std::string> path;
            std::string e = key_path(path);
            if (!e.empty())
                return e;
            skip_ws();
            if (!consume('='))
                return err("expected = after a key");
            skip_ws();
            Value v;
            e = value(v);
            if (!e.empty())
                return e;
            Value* tgt = cur;
            for (size_t k = 0; k + 1 < path.size(); k++) {
                Value& child = tgt->t[path[k]];
                if (child.kind == Value::NONE)
                    child.kind = Value::TABLE;
                if (child.kind != Value::TABLE)
                    return err("dotted key through a non-table");
                tgt = &child;
            }
            if (tgt->t.count(path.back()))
                return err("duplicate key " + path.back());
            tgt->t[path.back()] = std::move(v);
            if (!end_of_line())
                return err("unexpected characters after a value");
        }
    }

private:
    const std::string& s_;
    size_t pos_ = 0;
    int line_ = 1;

    bool eof() const { return pos_ >= s_.size(); }
    char peek() const { return s_[pos_]; }
    bool consume(char c)
    {
        if (!eof() && s_[pos_] == c) {
            pos_++;
            return true;
        }
        return false;
    }
    std::string err(const std::string& m) const { return "line " + std::to_string(line_) + ": " + m; }
    void skip_ws()
    {
        while (!eof() && (peek() == ' ' || peek() == '\t'))
            pos_++;
    }
    void skip_ws_and_comments(bool newlines)
    {
        while (!eof()) {
            const char c = peek();
            if (c == ' ' || c == '\t' || c == '\r')
                pos_++;
            else if (c == '\n' && newlines)
                pos_++, line_++;
            else if (c == '#')
                while (!eof() && peek() != '\n')
                    pos_++;
            else
                break;
        }
    }
    bool end_of_line()
    {
        skip_ws_and_comments(false);
        if (eof())
            return true;
        if (peek() == '\n') {
            pos_++, line_++;
            return true;
        }
        return false;
    }
    std::string key_path(std::vector<std::string>& out)
    {
        while (true) {
            skip_ws();
            std::string k;
            if (!eof() && (peek() == '"' || peek() == '\'')) {
                Value v;
                std::string e = str(v);
                if (!e.empty())
                    return e;
                k = v.s;
            } else {
                while (!eof() && (std::isalnum((unsigned char)peek()) || peek() == '_' || peek() == '-'))
                    k += s_[pos_++];
            }
            if (k.empty())
                return err("expected a key");
            out.push_back(k);
            skip_ws();
            if (!consume('.'))
                return "";
        }
    }
    std::string str(Value& v)
    {
        const char q = s_[pos_++];
        v.kind = Value::STRING;
        while (!eof() && peek() != q) {
            if (peek() == '\n')
                return err("unterminated string");
            if (q == '"' && peek() == '\\' && pos_ + 1 < s_.size()) {
                const char n = s_[pos_ + 1];
                pos_ += 2;
                v.s += n == 'n' ? '\n' : n == 't' ? '\t' : n; // \\ \" and friends
                continue;
            }
            v.s += s_[pos_++];
        }
        if (!consume(q))
            return err("unterminated string");
        return "";
    }
    std::string value(Value& v)
    {
        if (eof())
            return err("expected a value");
        const char c = peek();
        if (c == '"' || c == '\'')
            return str(v);
        if (c == '[') {
            pos_++;
            v.kind = Value::ARRAY;
            while (true) {
                skip_ws_and_comments(true);
                if (consume(']'))
                    return "";
                Value e;
                std::string er = value(e);
                if (!er.empty())
                    return er;
                v.a.push_back(std::move(e));
                skip_ws_and_comments(true);
                if (consume(','))
                    continue;
                if (consume(']'))
                    return "";
                return err("expected , or ] in an array");
            }
        }
        std::string tok;
        while (!eof() && !std::isspace((unsigned char)peek()) && peek() != ',' && peek() != ']' && peek() != '#')
            tok += s_[pos_++];
        if (tok == "true" || tok == "false") {
            v.kind = Value::BOOL, v.b = tok == "true";
            return "";
        }
        std::string num;
        for (char ch : tok)
            if (ch != '_')
                num += ch;
        if (num.empty())
            return err("expected a value");
        char* end = nullptr;
        if (num.find_first_of(".eE") == std::string::npos && num != "inf" && num != "nan" && num != "+inf" && num != "-inf") {
            v.i = std::strtoll(num.c_str(), &end, 0);
            v.kind = Value::INT;
        } else {
            v.f = std::strtod(num.c_str(), &end);
            v.kind = Value::FLOAT;
        }
        if (!end || *end != '\0')
            return err("cannot parse the value '" + tok + "'");
        return "";
    }
};

} // namespace mini_toml
