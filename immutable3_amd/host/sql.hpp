// sql.hpp -- SQLParser: the SQL subset of engine/src/main/scala/immutabledb/sql/SQLParser.scala:8-129,
// restated as a backtracking recursive-descent parser with the semantics of scala-parser-combinators'
// JavaTokenParsers: whitespace is skipped before every literal / regex, literals match as PREFIXES (no word
// boundary), alternatives are tried in order with backtracking, parseAll must consume the whole input.
//
//   query            := queryProjectAgg | queryProjectAggNoGroup | queryProject            (:13)
//   queryProject     := "select" repsep(ident, ",") "from" ident where limit               (:27-31, :45-47, :115-116)
//   queryProjectAgg  := "select" rep1sep(agg, ",") "from" ident where ["group" "by" rep1sep(ident, ",")]   (:15-25)
//   where            := opt("where" filter)                                                  (:49-54)
//   filter           := "(" repsep(filter,"and") ")" | "(" repsep(filter,"or") ")"
//                     | ident "=" value | ident "=" "'" value "'" | ident ">" value | ident "<" value   (:56-96)
//   limit            := opt("limit" digits)                                                  (:37)
//   ident = [\w#]+   value = [\w0-9#]+  (no sign, no decimal point)                          (:119-121)
#pragma once

#include <cctype>
#include <cstdlib>
#include <cstring>

#include "schema.hpp"

namespace immutabledb {

class SQLParser {
  public:
    static Query parseAll(const std::string &input) {
        SQLParser p(input);
        Query q;
        size_t end = 0;
        if (p.query(0, q, end)) {
            end = p.skipWs(end);
            if (end == input.size()) return q;
            throw Exception("[1." + std::to_string(end + 1) + "] failure: end of input expected\n\n" + input);
        }
        throw Exception("[1." + std::to_string(p.furthest_ + 1) + "] failure: " + p.expected_ + " expected\n\n" + input);
    }

  private:
    explicit SQLParser(const std::string &s) : s_(s) {}
    const std::string &s_;
    size_t furthest_ = 0;
    std::string expected_ = "`select'";

    size_t skipWs(size_t i) const {
        while (i < s_.size() && std::isspace((unsigned char)s_[i])) ++i;
        return i;
    }
    void note(size_t at, const std::string &what) {
        if (at >= furthest_) { furthest_ = at; expected_ = what; }
    }
    bool lit(size_t i, const char *word, size_t &out) {
        i = skipWs(i);
        const size_t n = std::strlen(word);
        if (s_.compare(i, n, word) == 0) { out = i + n; return true; }
        note(i, std::string("`") + word + "'");
        return false;
    }
    static bool wordChar(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '#'; }
    bool ident(size_t i, std::string &val, size_t &out) { // [\w#]+ (value has the same character class)
        i = skipWs(i);
        size_t j = i;
        while (j < s_.size() && wordChar(s_[j])) ++j;
        if (j == i) { note(i, "string matching regex `[\\w\\#]+'"); return false; }
        val = s_.substr(i, j - i);
        out = j;
        return true;
    }
    static double toDouble(const std::string &v) { // String.toDouble on [\w#]+ : NumberFormatException unless numeric
        char *end = nullptr;
        const double d = std::strtod(v.c_str(), &end);
        bool ok = !v.empty() && end == v.c_str() + v.size();
        // Java's parser accepts a trailing d/D/f/F and hex floats; strtod accepts "inf"/"nan"/hex too. Keep digits-only plus those.
        if (!ok && !v.empty() && (v.back() == 'd' || v.back() == 'D' || v.back() == 'f' || v.back() == 'F')) {
            const std::string w = v.substr(0, v.size() - 1);
            const double d2 = std::strtod(w.c_str(), &end);
            if (!w.empty() && end == w.c_str() + w.size()) return d2;
        }
        if (!ok) throw Exception("NumberFormatException: For input string: \"" + v + "\"");
        return d;
    }

    // ---- filters ----
    bool filterList(size_t i, const char *sep, bool isAnd, std::shared_ptr<SelectADT> &out, size_t &end) {
        size_t p;
        if (!lit(i, "(", p)) return false;
        std::vector<std::shared_ptr<SelectADT>> xs;
        std::shared_ptr<SelectADT> f;
        size_t q;
        if (filter(p, f, q)) { // repsep: zero or more
            xs.push_back(f);
            p = q;
            for (;;) {
                size_t r;
                if (!lit(p, sep, r)) break;
                if (!filter(r, f, q)) break; // repsep backtracks over the dangling separator
                xs.push_back(f);
                p = q;
            }
        }
        if (!lit(p, ")", q)) return false;
        if (xs.empty()) throw Exception("UnsupportedOperationException: tail of empty list"); // xs.tail on Nil (:64)
        std::shared_ptr<SelectADT> acc = xs[0];
        for (size_t k = 1; k < xs.size(); ++k) acc = isAnd ? SelectADT::mkAnd(acc, xs[k]) : SelectADT::mkOr(acc, xs[k]);
        out = acc;
        end = q;
        return true;
    }
    bool filter(size_t i, std::shared_ptr<SelectADT> &out, size_t &end) {
        if (filterList(i, "and", true, out, end)) return true;  // filterAnd
        if (filterList(i, "or", false, out, end)) return true;  // filterOr
        std::string f, v;
        size_t p, q, r, t;
        if (ident(i, f, p) && lit(p, "=", q) && ident(q, v, r)) { // filterEQ
            out = SelectADT::mkSelect(f, SelectCondition::eq(toDouble(v)));
            end = r;
            return true;
        }
        if (ident(i, f, p) && lit(p, "=", q) && lit(q, "'", r) && ident(r, v, t)) { // filterEQString
            size_t u;
            if (lit(t, "'", u)) {
                out = SelectADT::mkSelect(f, SelectCondition::match({v}));
                end = u;
                return true;
            }
        }
        if (ident(i, f, p) && lit(p, ">", q) && ident(q, v, r)) { // filterGT
            out = SelectADT::mkSelect(f, SelectCondition::gt(toDouble(v)));
            end = r;
            return true;
        }
        if (ident(i, f, p) && lit(p, "<", q) && ident(q, v, r)) { // filterLT
            out = SelectADT::mkSelect(f, SelectCondition::lt(toDouble(v)));
            end = r;
            return true;
        }
        return false;
    }
    bool where(size_t i, std::shared_ptr<SelectADT> &out, size_t &end) { // opt("where" ~> filter)
        size_t p, q;
        std::shared_ptr<SelectADT> f;
        if (lit(i, "where", p) && filter(p, f, q)) { out = f; end = q; return true; }
        out = SelectADT::noSelect();
        end = i;
        return true;
    }
    bool fromTable(size_t i, std::string &t, size_t &end) {
        size_t p;
        return lit(i, "from", p) && ident(p, t, end);
    }
    bool identList(size_t i, bool atLeastOne, std::vector<std::string> &out, size_t &end) {
        std::string f;
        size_t p;
        out.clear();
        if (!ident(i, f, p)) { end = i; return !atLeastOne; }
        out.push_back(f);
        for (;;) {
            size_t q, r;
            if (!lit(p, ",", q) || !ident(q, f, r)) break;
            out.push_back(f);
            p = r;
        }
        end = p;
        return true;
    }
    bool agg(size_t i, Aggregate &a, size_t &end) { // aggSumP | aggMinP | aggMaxP | aggCountP (:103-117)
        static const struct { const char *kw; Aggregate::Kind k; } kinds[] = {
            {"sum", Aggregate::Sum}, {"min", Aggregate::Min}, {"max", Aggregate::Max}, {"count", Aggregate::Count}};
        for (const auto &k : kinds) {
            size_t p, q, r, t;
            std::string name;
            if (lit(i, k.kw, p) && lit(p, "(", q) && ident(q, name, r) && lit(r, ")", t)) {
                a.kind = k.k;
                a.col = name;
                a.alias.clear();
                end = t;
                return true;
            }
        }
        return false;
    }
    bool selectProjectAgg(size_t i, ProjectADT &p, size_t &end) {
        size_t q;
        if (!lit(i, "select", q)) return false;
        Aggregate a;
        size_t r;
        if (!agg(q, a, r)) return false;
        p = ProjectADT();
        p.kind = ProjectADT::ProjectAgg;
        p.aggs.push_back(a);
        for (;;) {
            size_t c, t;
            if (!lit(r, ",", c) || !agg(c, a, t)) break;
            p.aggs.push_back(a);
            r = t;
        }
        end = r;
        return true;
    }
    bool query(size_t i, Query &out, size_t &end) {
        {   // queryProjectAgg: selectProjectAgg ~ fromTable ~ where ~ groupBy
            ProjectADT p;
            size_t a, b, c, d, e, f;
            std::string t;
            std::shared_ptr<SelectADT> w;
            if (selectProjectAgg(i, p, a) && fromTable(a, t, b) && where(b, w, c)) {
                std::vector<std::string> g;
                if (lit(c, "group", d) && lit(d, "by", e) && identList(e, true, g, f)) {
                    p.groupBy = g;
                    out = Query{t, w, p};
                    end = f;
                    // parseAll: this alternative only wins if it consumes everything; otherwise Scala's `|`
                    // has already committed to it (first success), and parseAll then fails on the leftovers.
                    return true;
                }
                // queryProjectAggNoGroup
                out = Query{t, w, p};
                end = c;
                return true;
            }
        }
        {   // queryProject: selectProject ~ fromTable ~ where ~ limit
            size_t a, b, c, d;
            std::vector<std::string> cols;
            std::string t;
            std::shared_ptr<SelectADT> w;
            if (lit(i, "select", a) && identList(a, false, cols, b) && fromTable(b, t, c) && where(c, w, d)) {
                ProjectADT p;
                p.kind = ProjectADT::Project;
                p.cols = cols;
                size_t e;
                if (lit(d, "limit", e)) {
                    size_t k = skipWs(e), j = k;
                    while (j < s_.size() && std::isdigit((unsigned char)s_[j])) ++j;
                    if (j > k) {
                        p.limit = std::atoi(s_.substr(k, j - k).c_str());
                        d = j;
                    }
                }
                out = Query{t, w, p};
                end = d;
                return true;
            }
        }
        return false;
    }
};

} // namespace immutabledb
