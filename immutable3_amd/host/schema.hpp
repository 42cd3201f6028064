// schema.hpp -- Column / Table / TableIO / Row, mirroring core/src/main/scala/immutabledb/{Column,Table,Record}.scala
// and the Query ADT of core/src/main/scala/immutabledb/Query.scala (same names, same fields).
#pragma once

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "json.hpp"

namespace immutabledb {

// `throw new Exception(msg)` of the reference
struct Exception : std::runtime_error {
    explicit Exception(const std::string &m) : std::runtime_error(m) {}
};

// ---- CodecType (codec/Codec.scala:21-24) / ColumnType (Column.scala:13-16) ----
// SNAPPY_* are this library's extension (include/imm3.h): blocks as SnappyCodec.encode writes them
// (codec/SnappyCodec.scala:15-43); the reference has no CodecType for them and cannot read them.
enum class CodecType : int { PFOR_INT = 0, DENSE_INT = 1, DENSE_TINYINT = 2, DENSE_STRING = 3, SNAPPY_INT = 16, SNAPPY_TINYINT = 17, SNAPPY_STRING = 18 };
inline bool isSnappy(CodecType c) { return (int)c >= 16; }
enum class ColumnType : int { INT = 0, TINYINT = 1, STRING = 2 };

inline const char *toString(CodecType c) {
    static const char *n[] = {"PFOR_INT", "DENSE_INT", "DENSE_TINYINT", "DENSE_STRING"};
    static const char *x[] = {"SNAPPY_INT", "SNAPPY_TINYINT", "SNAPPY_STRING"};
    return isSnappy(c) ? x[(int)c - 16] : n[(int)c];
}
inline const char *toString(ColumnType c) {
    static const char *n[] = {"INT", "TINYINT", "STRING"};
    return n[(int)c];
}
inline CodecType codecWithName(const std::string &s) {
    for (int i = 0; i < 4; ++i)
        if (s == toString((CodecType)i)) return (CodecType)i;
    for (int i = 16; i < 19; ++i)
        if (s == toString((CodecType)i)) return (CodecType)i;
    throw Exception("No value found for '" + s + "'"); // Enumeration.withName
}
inline ColumnType columnTypeWithName(const std::string &s) {
    for (int i = 0; i < 3; ++i)
        if (s == toString((ColumnType)i)) return (ColumnType)i;
    throw Exception("No value found for '" + s + "'");
}

// ---- Column (Column.scala:18-63) ----
struct Column {
    std::string name;
    ColumnType columnType = ColumnType::INT;
    CodecType codec = CodecType::DENSE_INT;
    std::vector<std::pair<std::string, std::string>> dtypeAttrs; // Map[String, String]

    static Column make(const std::string &name, CodecType codec, std::vector<std::pair<std::string, std::string>> attrs = {}) {
        Column c;
        c.name = name;
        c.codec = codec;
        c.dtypeAttrs = std::move(attrs);
        switch (codec) {
        case CodecType::DENSE_INT:
        case CodecType::SNAPPY_INT:
        case CodecType::PFOR_INT: c.columnType = ColumnType::INT; break;
        case CodecType::SNAPPY_TINYINT:
        case CodecType::DENSE_TINYINT: c.columnType = ColumnType::TINYINT; break;
        case CodecType::SNAPPY_STRING:
        case CodecType::DENSE_STRING: c.columnType = ColumnType::STRING; break;
        }
        return c;
    }
    std::string attr(const std::string &k) const {
        for (const auto &kv : dtypeAttrs)
            if (kv.first == k) return kv.second;
        throw Exception("key not found: " + k);
    }
    // dtype.size of Column.getCodec(column) (Column.scala:57-63; DataType.scala:34,54,64)
    int width() const {
        switch (codec) {
        case CodecType::DENSE_INT:
        case CodecType::SNAPPY_INT:
        case CodecType::PFOR_INT: return 4;
        case CodecType::SNAPPY_TINYINT:
        case CodecType::DENSE_TINYINT: return 1;
        case CodecType::SNAPPY_STRING:
        case CodecType::DENSE_STRING: return std::stoi(attr("size"));
        }
        throw Exception("");
    }
    bool operator==(const Column &o) const { return name == o.name && columnType == o.columnType && codec == o.codec && dtypeAttrs == o.dtypeAttrs; }

    json::Value toJsonValue() const { // Column.scala:21-29
        json::Value v = json::Value::object();
        v.put("name", json::Value::string(name));
        v.put("columnType", json::Value::string(toString(columnType)));
        v.put("codec", json::Value::string(toString(codec)));
        json::Value a = json::Value::object();
        for (const auto &kv : dtypeAttrs) a.put(kv.first, json::Value::string(kv.second));
        v.put("dtypeAttrs", a);
        return v;
    }
    static Column fromJsonValue(const json::Value &j) { // Column.scala:31-38
        Column c;
        c.name = j.at("name").str;
        c.columnType = columnTypeWithName(j.at("columnType").str);
        c.codec = codecWithName(j.at("codec").str);
        for (const auto &kv : j.at("dtypeAttrs").obj) c.dtypeAttrs.emplace_back(kv.first, kv.second.str);
        return c;
    }
};

// ---- Table / TableIO (Table.scala:9-66) ----
struct Table {
    std::string name;
    std::vector<Column> columns;
    int blockSize = 1024;

    const Column &getColumn(const std::string &colName) const {
        for (const auto &c : columns)
            if (c.name == colName) return c;
        throw Exception("Column " + colName + " does not exist in table " + name);
    }
    int columnIndex(const std::string &colName) const {
        for (size_t i = 0; i < columns.size(); ++i)
            if (columns[i].name == colName) return (int)i;
        throw Exception("Column " + colName + " does not exist in table " + name);
    }
};

inline std::string readFile(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Exception(path + " (No such file or directory)");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

inline void mkdirs(const std::string &path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty()) ::mkdir(cur.c_str(), 0777);
        }
        if (i < path.size()) cur += path[i];
    }
}

inline std::vector<std::string> listDir(const std::string &path, bool dirsOnly) {
    std::vector<std::string> out;
    DIR *d = ::opendir(path.c_str());
    if (!d) return out;
    while (struct dirent *e = ::readdir(d)) {
        const std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        struct stat st {};
        if (::stat((path + "/" + n).c_str(), &st) != 0) continue;
        if (dirsOnly ? S_ISDIR(st.st_mode) : S_ISREG(st.st_mode)) out.push_back(n);
    }
    ::closedir(d);
    return out;
}

struct TableIO {
    static constexpr const char *fileName = "_table.meta";

    static json::Value toJsonValue(const Table &t) { // Table.scala:29-35
        json::Value v = json::Value::object();
        v.put("name", json::Value::string(t.name));
        json::Value cols = json::Value::array();
        for (const auto &c : t.columns) cols.arr.push_back(c.toJsonValue());
        v.put("columns", cols);
        v.put("blockSize", json::Value::number(t.blockSize));
        return v;
    }
    static Table fromJsonValue(const json::Value &j) { // Table.scala:37-43
        Table t;
        t.name = j.at("name").str;
        for (const auto &c : j.at("columns").arr) t.columns.push_back(Column::fromJsonValue(c));
        t.blockSize = (int)j.at("blockSize").num;
        return t;
    }
    static Table load(const std::string &dataDir, const std::string &tableName) {
        return fromJsonValue(json::parse(readFile(dataDir + "/" + tableName + "/" + fileName)));
    }
    static void store(const std::string &dataDir, const Table &t) {
        const std::string path = dataDir + "/" + t.name;
        mkdirs(path);
        std::ofstream f(path + "/" + fileName, std::ios::binary | std::ios::trunc);
        f << json::dump(toJsonValue(t));
    }
    static void clear(const std::string &dataDir, const Table &t) { // Table.scala:61-66: delete the table's files
        const std::string path = dataDir + "/" + t.name;
        for (const auto &f : listDir(path, false)) ::unlink((path + "/" + f).c_str());
    }
};

// ---- Row (Record.scala:7-14): boxed Int / Byte / String values ----
struct Value {
    enum Kind { Int, Byte, String } kind = Int;
    int32_t i = 0;
    std::string s;
    static Value ofInt(int32_t v) { Value x; x.kind = Int; x.i = v; return x; }
    static Value ofByte(int8_t v) { Value x; x.kind = Byte; x.i = v; return x; }
    static Value ofString(std::string v) { Value x; x.kind = String; x.s = std::move(v); return x; }
    std::string toString() const { return kind == String ? s : std::to_string(i); }
    bool operator==(const Value &o) const { return kind == o.kind && i == o.i && s == o.s; }
};

struct Row {
    std::vector<Value> xs;
    static Row fromSeq(std::vector<Value> v) { Row r; r.xs = std::move(v); return r; }
    int8_t getByte(size_t idx) const { return (int8_t)xs.at(idx).i; }
    int32_t getInt(size_t idx) const { return xs.at(idx).i; }
    const std::string &getString(size_t idx) const { return xs.at(idx).s; }
    std::string toString() const { // xs.mkString("Row(", ",", ")")
        std::string out = "Row(";
        for (size_t k = 0; k < xs.size(); ++k) {
            if (k) out += ",";
            out += xs[k].toString();
        }
        return out + ")";
    }
};

// ---- Query ADT (Query.scala:3-46) ----
struct SelectCondition {
    enum Kind { Match, NotMatch, EQ, GT, LT, NoOp } kind = NoOp; // same order as IMM3_MATCH .. IMM3_NOOP
    double value = 0;                 // EQ / GT / LT
    std::vector<std::string> values;  // Match / NotMatch
    static SelectCondition match(std::vector<std::string> v) { SelectCondition c; c.kind = Match; c.values = std::move(v); return c; }
    static SelectCondition notMatch(std::vector<std::string> v) { SelectCondition c; c.kind = NotMatch; c.values = std::move(v); return c; }
    static SelectCondition eq(double d) { SelectCondition c; c.kind = EQ; c.value = d; return c; }
    static SelectCondition gt(double d) { SelectCondition c; c.kind = GT; c.value = d; return c; }
    static SelectCondition lt(double d) { SelectCondition c; c.kind = LT; c.value = d; return c; }
    std::string toString() const {
        auto list = [&]() { std::string s = "List("; for (size_t i = 0; i < values.size(); ++i) { if (i) s += ", "; s += values[i]; } return s + ")"; };
        std::ostringstream d;
        d << value;
        switch (kind) {
        case Match: return "Match(" + list() + ")";
        case NotMatch: return "NotMatch(" + list() + ")";
        case EQ: return "EQ(" + d.str() + ")";
        case GT: return "GT(" + d.str() + ")";
        case LT: return "LT(" + d.str() + ")";
        default: return "NoOp";
        }
    }
};

struct SelectADT {
    enum Kind { And, Or, Select, NoSelect } kind = NoSelect;
    std::shared_ptr<SelectADT> op1, op2; // And / Or
    std::string col;                     // Select
    SelectCondition cond;
    static std::shared_ptr<SelectADT> mkAnd(std::shared_ptr<SelectADT> a, std::shared_ptr<SelectADT> b) { auto s = std::make_shared<SelectADT>(); s->kind = And; s->op1 = a; s->op2 = b; return s; }
    static std::shared_ptr<SelectADT> mkOr(std::shared_ptr<SelectADT> a, std::shared_ptr<SelectADT> b) { auto s = std::make_shared<SelectADT>(); s->kind = Or; s->op1 = a; s->op2 = b; return s; }
    static std::shared_ptr<SelectADT> mkSelect(const std::string &c, SelectCondition cond) { auto s = std::make_shared<SelectADT>(); s->kind = Select; s->col = c; s->cond = std::move(cond); return s; }
    static std::shared_ptr<SelectADT> noSelect() { return std::make_shared<SelectADT>(); }
};

struct Aggregate { // Query.scala:17-26
    enum Kind { Sum, Avg, Min, Max, Count } kind = Count;
    std::string col;
    std::string alias; // empty = None
};

struct ProjectADT {
    enum Kind { Project, ProjectAgg, NoProject } kind = NoProject;
    std::vector<std::string> cols; // Project
    int limit = 0;
    std::vector<Aggregate> aggs;   // ProjectAgg
    std::vector<std::string> groupBy;
};

struct Query {
    std::string table;
    std::shared_ptr<SelectADT> select;
    ProjectADT project;
};

} // namespace immutabledb
