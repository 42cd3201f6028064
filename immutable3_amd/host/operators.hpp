// operators.hpp -- GPU-backed Operator / ScanOp / SelectOp / ProjectOp / Engine: the C++ host-side mirror of the
// reference's operator package above the C ABI (include/imm3.h).  Same names, argument meaning and error texts:
//
//   engine/src/main/scala/immutabledb/engine/operator/Operator.scala:14-28   Operator / ColumnVectorOperator / ProjectionOperator
//   engine/src/main/scala/immutabledb/engine/operator/Scan.scala:10-73       ScanOp, mkScanOp
//   engine/src/main/scala/immutabledb/engine/operator/Select.scala:5-165     SelectOp, mkSelectOp
//   engine/src/main/scala/immutabledb/engine/operator/Project.scala:8-81     ProjectOp, mkProjectOp
//   engine/src/main/scala/immutabledb/engine/Engine.scala:85-128,158-262     getColumns / resolveSelectOps / execute
//   core/src/main/scala/immutabledb/DataVector.scala:15-48                   ColumnVectorBatch family
//
// Operators are plan builders: the first iterator() call fuses ScanOp -> SelectOp* (-> ProjectOp) of one segment
// into one imm3_query.  Everything that touches data goes through the C ABI; there is no CPU evaluation path.
#pragma once

#include <functional>
#include <memory>

#include "../../include/imm3.h"
#include "storage.hpp"
#include "scala_sets.hpp"

namespace immutabledb {

inline void imm3Check(int rc) {
    if (rc != IMM3_OK) throw Exception(imm3_last_error());
}

// ---- vectors (DataVector.scala) ----
struct BitSet { // scala.collection.mutable.BitSet over the GPU's words: bit i <-> word i>>6, bit i&63
    std::vector<uint64_t> words;
    bool contains(int i) const { return (size_t)(i >> 6) < words.size() && ((words[(size_t)(i >> 6)] >> (i & 63)) & 1ULL); }
    int size() const { int c = 0; for (uint64_t w : words) c += __builtin_popcountll(w); return c; }
    bool isEmpty() const { for (uint64_t w : words) if (w) return false; return true; }
    std::vector<int> toList() const {
        std::vector<int> out;
        for (size_t k = 0; k < words.size(); ++k)
            for (uint64_t w = words[k]; w; w &= w - 1) out.push_back((int)(k * 64 + (size_t)__builtin_ctzll(w)));
        return out;
    }
};

struct ColumnVector { // Int / TinyInt / String column vector: a typed VIEW of the block (DENSE decode is a reinterpretation)
    ColumnType type = ColumnType::INT;
    const uint8_t *data = nullptr;
    int width = 4;
    int n = 0;
    std::shared_ptr<std::vector<uint8_t>> owner; // PFOR_INT: the column decoded by the GPU (data points into it)
    Value value(int pos) const {
        switch (type) {
        case ColumnType::INT: { int32_t v; std::memcpy(&v, data + (size_t)pos * 4, 4); return Value::ofInt(v); }
        case ColumnType::TINYINT: return Value::ofByte((int8_t)data[pos]);
        default: return Value::ofString(std::string((const char *)data + (size_t)pos * (size_t)width, (size_t)width)); // new String(bytes)
        }
    }
};

struct ColumnVectorBatch { // FilledColumnVectorBatch (DataVector.scala:24-31)
    int oid = 0;
    int size = 0;
    std::vector<ColumnVector> columnVectors;
    std::vector<Column> columns;
    BitSet selected;
    bool selectedInUse = true;
};

// ---- pull iterators (scala Iterator) ----
template <typename A>
struct Iterator {
    virtual ~Iterator() = default;
    virtual bool hasNext() = 0;
    virtual A next() = 0;
};

template <typename A>
struct Operator { // Operator.scala:14-16
    virtual ~Operator() = default;
    virtual std::unique_ptr<Iterator<A>> iterator() = 0;
};
using ColumnVectorOperator = Operator<ColumnVectorBatch>; // Operator.scala:18-20
using ProjectionOperator = Operator<Row>;                 // Operator.scala:26-28

template <typename A>
struct VectorIterator : Iterator<A> {
    std::vector<A> items;
    size_t pos = 0;
    bool hasNext() override { return pos < items.size(); }
    A next() override { return std::move(items[pos++]); }
};

// ---- device-resident SegmentManager ----
class GpuSegmentManager {
  public:
    explicit GpuSegmentManager(const SegmentManager &sm, int device = 0) : sm(sm) { imm3Check(imm3_ctx_create(device, nullptr, &ctx_)); }
    ~GpuSegmentManager() {
        for (auto &kv : tables_) imm3_table_destroy(kv.second);
        for (auto &kv : segs_) imm3_segment_destroy(kv.second);
        imm3_ctx_destroy(ctx_);
    }
    // every segment of the table as ONE scan unit (imm3_table), or nullptr when the table cannot take the
    // single-launch path (ragged segments): callers then fall back to one pipeline per segment
    imm3_table *deviceTable(const std::string &tableName) {
        auto it = tables_.find(tableName);
        if (it != tables_.end()) return it->second;
        std::vector<const imm3_segment *> segs;
        const int n = sm.getTableSegmentCount(tableName);
        for (int s = 0; s < n; ++s) segs.push_back(deviceSegment(tableName, s));
        imm3_table *t = nullptr;
        if (!segs.empty()) {
            const int rc = imm3_table_create(ctx_, segs.data(), (int32_t)segs.size(), &t);
            if (rc != IMM3_OK && rc != IMM3_ERR_LAYOUT) throw Exception(imm3_last_error());
            if (rc != IMM3_OK) t = nullptr;
        }
        tables_[tableName] = t;
        return t;
    }
    GpuSegmentManager(const GpuSegmentManager &) = delete;
    GpuSegmentManager &operator=(const GpuSegmentManager &) = delete;

    imm3_ctx *ctx() const { return ctx_; }
    // all columns of one segment id, staged into HBM once (SegmentManager keeps its mmaps the same way)
    imm3_segment *deviceSegment(const std::string &tableName, int segIdx) {
        const auto key = std::make_pair(tableName, segIdx);
        auto it = segs_.find(key);
        if (it != segs_.end()) return it->second;
        const Table &t = sm.getTable(tableName);
        std::vector<imm3_column> cols(t.columns.size());
        for (size_t i = 0; i < t.columns.size(); ++i) {
            const std::string k = tableName + "." + t.columns[i].name;
            const MappedFile &f = sm.segments.at(k).at((size_t)segIdx);
            const SegmentMeta &m = sm.segmentsMeta.at(k).at((size_t)segIdx);
            cols[i].codec = (int32_t)t.columns[i].codec;
            cols[i].width = t.columns[i].width();
            cols[i].dat = f.data;
            cols[i].dat_bytes = f.size;
            cols[i].block_offsets = m.blockOffsets.data();
            cols[i].n_offsets = (int32_t)m.blockOffsets.size();
        }
        imm3_segment *seg = nullptr;
        imm3Check(imm3_segment_create(ctx_, cols.data(), (int32_t)cols.size(), &seg));
        segs_[key] = seg;
        return seg;
    }
    const SegmentManager &sm;

  private:
    imm3_ctx *ctx_ = nullptr;
    std::map<std::pair<std::string, int>, imm3_segment *> segs_;
    std::map<std::string, imm3_table *> tables_;
};

struct Leaf { std::string col; SelectCondition cond; };

// RAII imm3_query
struct QueryHandle {
    imm3_query *q = nullptr;
    ~QueryHandle() { if (q) imm3_query_destroy(q); }
};

// ---- ScanOp (Scan.scala:17) ----
class ScanOp : public ColumnVectorOperator {
  public:
    ScanOp(GpuSegmentManager &sm, int segIdx, const std::string &tableName, std::vector<Column> cols)
        : sm_(sm), segIdx_(segIdx), tableName_(tableName), cols_(std::move(cols)) {}
    static std::function<std::shared_ptr<ScanOp>(const std::vector<Column> &, int)> mkScanOp(GpuSegmentManager &sm, const std::string &tableName) {
        return [&sm, tableName](const std::vector<Column> &cols, int segIdx) { return std::make_shared<ScanOp>(sm, segIdx, tableName, cols); };
    }
    const Table &table() const { return sm_.sm.getTable(tableName_); }
    GpuSegmentManager &manager() const { return sm_; }

    // builds the fused query: leaves in application order, optional projection
    void makeQuery(const std::vector<Leaf> &leaves, const std::vector<std::string> &projNames, int limit, QueryHandle &h) const {
        const Table &t = table();
        std::vector<int32_t> used;
        for (const auto &c : cols_) used.push_back(t.columnIndex(c.name));
        auto usedIndex = [&](const std::string &name) -> int32_t {
            for (size_t i = 0; i < cols_.size(); ++i)
                if (cols_[i].name == name) return (int32_t)i; // `.filter(_.name == col).head`, Select.scala:60
            throw Exception("NoSuchElementException: next on empty iterator");
        };
        std::vector<imm3_select> sels(leaves.size());
        std::vector<std::string> blobs(leaves.size());
        std::vector<std::vector<int32_t>> lens(leaves.size());
        for (size_t i = 0; i < leaves.size(); ++i) {
            sels[i] = imm3_select{};
            sels[i].column = usedIndex(leaves[i].col);
            sels[i].cond = (int32_t)leaves[i].cond.kind;
            sels[i].value = leaves[i].cond.value;
            for (const auto &v : leaves[i].cond.values) { blobs[i] += v; lens[i].push_back((int32_t)v.size()); }
            sels[i].match_bytes = (const uint8_t *)blobs[i].data();
            sels[i].match_lens = lens[i].data();
            sels[i].n_match = (int32_t)lens[i].size();
        }
        std::vector<int32_t> proj;
        for (const auto &n : projNames) {
            bool found = false;
            for (size_t i = 0; i < cols_.size() && !found; ++i)
                if (cols_[i].name == n) { proj.push_back((int32_t)i); found = true; }
            if (!found) throw Exception("NoSuchElementException: key not found: " + n); // vecCols(col), Project.scala:56
        }
        imm3Check(imm3_query_create(sm_.ctx(), sm_.deviceSegment(tableName_, segIdx_), used.data(), (int32_t)used.size(),
                                    sels.data(), (int32_t)sels.size(), proj.data(), (int32_t)proj.size(), limit, t.blockSize, &h.q));
    }

    void makeAggQuery(const std::vector<Leaf> &leaves, const std::vector<int32_t> &group, const std::vector<imm3_aggregate> &aggs, QueryHandle &h) const {
        const Table &t = table();
        std::vector<int32_t> used;
        for (const auto &c : cols_) used.push_back(t.columnIndex(c.name));
        std::vector<imm3_select> sels(leaves.size());
        std::vector<std::string> blobs(leaves.size());
        std::vector<std::vector<int32_t>> lens(leaves.size());
        for (size_t i = 0; i < leaves.size(); ++i) {
            sels[i] = imm3_select{};
            sels[i].column = -1;
            for (size_t k = 0; k < cols_.size(); ++k) if (cols_[k].name == leaves[i].col) { sels[i].column = (int32_t)k; break; }
            if (sels[i].column < 0) throw Exception("NoSuchElementException: next on empty iterator");
            sels[i].cond = (int32_t)leaves[i].cond.kind;
            sels[i].value = leaves[i].cond.value;
            for (const auto &v : leaves[i].cond.values) { blobs[i] += v; lens[i].push_back((int32_t)v.size()); }
            sels[i].match_bytes = (const uint8_t *)blobs[i].data();
            sels[i].match_lens = lens[i].data();
            sels[i].n_match = (int32_t)lens[i].size();
        }
        imm3Check(imm3_query_create_agg(sm_.ctx(), sm_.deviceSegment(tableName_, segIdx_), used.data(), (int32_t)used.size(),
                                        sels.data(), (int32_t)sels.size(), group.data(), (int32_t)group.size(),
                                        aggs.data(), (int32_t)aggs.size(), t.blockSize, &h.q));
    }

    std::unique_ptr<Iterator<ColumnVectorBatch>> batches(const std::vector<Leaf> &leaves) const {
        QueryHandle h;
        makeQuery(leaves, {}, 0, h);
        imm3Check(imm3_query_run_select(h.q));
        int32_t nb = 0;
        int64_t nwords = 0, nrows = 0;
        imm3Check(imm3_query_layout(h.q, &nb, &nwords, &nrows));
        std::vector<int32_t> size((size_t)nb), oid((size_t)nb);
        std::vector<int64_t> woff((size_t)nb);
        imm3Check(imm3_query_batches(h.q, size.data(), oid.data(), woff.data()));
        std::vector<uint64_t> words((size_t)nwords);
        imm3Check(imm3_query_bitmap(h.q, words.data(), nwords));
        auto it = std::make_unique<VectorIterator<ColumnVectorBatch>>();
        // PFOR_INT / snappy columns: the vectors of the batches are the GPU's decode of the segment (one projection, no predicate)
        std::vector<std::shared_ptr<std::vector<uint8_t>>> decoded(cols_.size());
        for (size_t ci = 0; ci < cols_.size(); ++ci) {
            if ((cols_[ci].codec != CodecType::PFOR_INT && !isSnappy(cols_[ci].codec)) || nrows == 0) continue;
            QueryHandle d;
            const Table &t = table();
            const int32_t used = t.columnIndex(cols_[ci].name), proj = 0;
            imm3Check(imm3_query_create(sm_.ctx(), sm_.deviceSegment(tableName_, segIdx_), &used, 1, nullptr, 0, &proj, 1, 0, t.blockSize, &d.q));
            imm3Check(imm3_query_run(d.q));
            uint64_t n = 0;
            imm3Check(imm3_query_row_count(d.q, &n));
            decoded[ci] = std::make_shared<std::vector<uint8_t>>((size_t)n * (size_t)cols_[ci].width());
            void *outp = decoded[ci]->data();
            imm3Check(imm3_query_fetch_rows(d.q, nullptr, &outp, n));
        }
        int64_t row = 0;
        for (int32_t k = 0; k < nb; ++k) {
            ColumnVectorBatch b;
            b.oid = oid[(size_t)k];
            b.size = size[(size_t)k];
            b.columns = cols_;
            const int64_t nw = (b.size + 63) / 64;
            b.selected.words.assign(words.begin() + woff[(size_t)k], words.begin() + woff[(size_t)k] + nw);
            b.selectedInUse = leaves.empty() ? true : !b.selected.isEmpty(); // Select.scala:44-47
            for (size_t ci = 0; ci < cols_.size(); ++ci) {
                const Column &c = cols_[ci];
                const MappedFile &f = sm_.sm.segments.at(tableName_ + "." + c.name).at((size_t)segIdx_);
                ColumnVector v;
                v.type = c.columnType;
                v.width = c.width();
                v.owner = decoded[ci];
                v.data = (decoded[ci] ? decoded[ci]->data() : f.data) + (size_t)row * (size_t)v.width;
                v.n = b.size;
                b.columnVectors.push_back(v);
            }
            row += b.size;
            it->items.push_back(std::move(b));
        }
        return it;
    }
    std::unique_ptr<Iterator<ColumnVectorBatch>> iterator() override { return batches({}); }
    const std::vector<Column> &cols() const { return cols_; }

  private:
    GpuSegmentManager &sm_;
    int segIdx_;
    std::string tableName_;
    std::vector<Column> cols_;
};

// ---- SelectOp (Select.scala:14) ----
class SelectOp : public ColumnVectorOperator {
  public:
    SelectOp(const std::string &col, SelectCondition cond, std::shared_ptr<ColumnVectorOperator> op) : col_(col), cond_(std::move(cond)), op_(std::move(op)) {}
    static std::function<std::shared_ptr<ColumnVectorOperator>(std::shared_ptr<ColumnVectorOperator>)> mkSelectOp(const std::string &col, const SelectCondition &cond) {
        return [col, cond](std::shared_ptr<ColumnVectorOperator> op) { return std::make_shared<SelectOp>(col, cond, op); };
    }
    // (ScanOp, leaves in application order: innermost SelectOp first)
    std::shared_ptr<ScanOp> chain(std::vector<Leaf> &leaves) const {
        std::shared_ptr<ScanOp> scan;
        if (auto inner = std::dynamic_pointer_cast<SelectOp>(op_)) scan = inner->chain(leaves);
        else if (auto s = std::dynamic_pointer_cast<ScanOp>(op_)) scan = s;
        else throw Exception("SelectOp chain must end in a ScanOp for the fused GPU path");
        leaves.push_back(Leaf{col_, cond_});
        return scan;
    }
    static void checkConditions(const std::vector<Leaf> &leaves) {
        for (const auto &l : leaves)
            if (l.cond.kind == SelectCondition::NotMatch || l.cond.kind == SelectCondition::NoOp)
                throw Exception("Unsupported condition: " + l.cond.toString()); // Select.scala:22
    }
    std::unique_ptr<Iterator<ColumnVectorBatch>> iterator() override {
        std::vector<Leaf> leaves;
        auto scan = chain(leaves);
        checkConditions(leaves);
        return scan->batches(leaves);
    }

  private:
    std::string col_;
    SelectCondition cond_;
    std::shared_ptr<ColumnVectorOperator> op_;
};

// ---- ProjectOp (Project.scala:17) ----
class ProjectOp : public ProjectionOperator {
  public:
    ProjectOp(std::vector<std::string> cols, std::shared_ptr<ColumnVectorOperator> op, int limit = 0) : cols_(std::move(cols)), op_(std::move(op)), limit_(limit) {}
    static std::function<std::shared_ptr<ProjectOp>(std::shared_ptr<ColumnVectorOperator>)> mkProjectOp(const std::vector<std::string> &cols, int limit = 0) {
        return [cols, limit](std::shared_ptr<ColumnVectorOperator> op) { return std::make_shared<ProjectOp>(cols, op, limit); };
    }
    std::unique_ptr<Iterator<Row>> iterator() override {
        std::vector<Leaf> leaves;
        std::shared_ptr<ScanOp> scan;
        if (auto sel = std::dynamic_pointer_cast<SelectOp>(op_)) scan = sel->chain(leaves);
        else scan = std::dynamic_pointer_cast<ScanOp>(op_);
        if (!scan) return hostIterator();
        SelectOp::checkConditions(leaves);
        QueryHandle h;
        scan->makeQuery(leaves, cols_, limit_, h);
        imm3Check(imm3_query_run(h.q));
        uint64_t n = 0;
        imm3Check(imm3_query_row_count(h.q, &n));
        std::vector<Column> pcols;
        for (const auto &name : cols_)
            for (const auto &c : scan->cols())
                if (c.name == name) { pcols.push_back(c); break; }
        std::vector<std::vector<uint8_t>> bufs(pcols.size());
        std::vector<void *> ptrs(pcols.size());
        for (size_t j = 0; j < pcols.size(); ++j) {
            bufs[j].resize((size_t)std::max<uint64_t>(n, 1) * (size_t)pcols[j].width());
            ptrs[j] = bufs[j].data();
        }
        imm3Check(imm3_query_fetch_rows(h.q, nullptr, ptrs.data(), n));
        auto it = std::make_unique<VectorIterator<Row>>();
        it->items.reserve((size_t)n);
        for (uint64_t i = 0; i < n; ++i) {
            std::vector<Value> xs;
            for (size_t j = 0; j < pcols.size(); ++j) {
                ColumnVector v;
                v.type = pcols[j].columnType;
                v.width = pcols[j].width();
                v.data = bufs[j].data();
                xs.push_back(v.value((int)i));
            }
            it->items.push_back(Row::fromSeq(std::move(xs)));
        }
        return it;
    }

  private:
    // ProjectIterator over batches from a non-fusable upstream (Project.scala:37-80); empty batches are skipped
    // (the reference faults on them).
    std::unique_ptr<Iterator<Row>> hostIterator() {
        auto it = std::make_unique<VectorIterator<Row>>();
        auto in = op_->iterator();
        int total = 0;
        while (in->hasNext() && !(limit_ > 0 && total >= limit_)) {
            ColumnVectorBatch vec = in->next();
            std::vector<int> vecCols;
            for (const auto &name : cols_) {
                int idx = -1;
                for (size_t i = 0; i < vec.columns.size(); ++i)
                    if (vec.columns[i].name == name) { idx = (int)i; break; }
                if (idx < 0) throw Exception("NoSuchElementException: key not found: " + name);
                vecCols.push_back(idx);
            }
            for (int pos : vec.selected.toList()) {
                if (limit_ > 0 && total >= limit_) break;
                std::vector<Value> xs;
                for (int ci : vecCols) xs.push_back(vec.columnVectors[(size_t)ci].value(pos));
                it->items.push_back(Row::fromSeq(std::move(xs)));
                ++total;
            }
        }
        return it;
    }
    std::vector<std::string> cols_;
    std::shared_ptr<ColumnVectorOperator> op_;
    int limit_;
};


// ---- aggregation (engine/.../operator/ProjectAggregate.scala:11-227, ProjectAggregateQueue.scala:9-55) ----
inline std::string javaDoubleToString(double v) { // Double.toString for the integral values the aggregators hold
    const long long iv = (long long)v;
    if (iv > -10000000LL && iv < 10000000LL) return std::to_string(iv) + ".0";
    std::string digits = std::to_string(iv < 0 ? -iv : iv);
    std::string frac = digits.substr(1);
    while (!frac.empty() && frac.back() == '0') frac.pop_back();
    if (frac.empty()) frac = "0";
    return std::string(iv < 0 ? "-" : "") + digits[0] + "." + frac + "E" + std::to_string(digits.size() - 1);
}

struct Aggregator { // CountAggr / MaxDoubleAggr / MinDoubleAggr / MaxStringAggr, folded into one tagged value
    enum Kind { CountAggr, MaxDoubleAggr, MinDoubleAggr, MaxStringAggr } kind = CountAggr;
    std::string col, alias;
    long long counter = 0;
    double dvalue = 0;
    std::string svalue;
    static Aggregator make(Kind k, const std::string &col, const std::string &alias) {
        Aggregator a;
        a.kind = k;
        a.col = col;
        a.alias = alias;
        a.dvalue = k == MaxDoubleAggr ? -1.7976931348623157e308 : 1.7976931348623157e308; // Double.MinValue / MaxValue
        return a;
    }
    int abiKind() const { return kind == CountAggr ? IMM3_AGG_COUNT : (kind == MinDoubleAggr ? IMM3_AGG_MIN : IMM3_AGG_MAX); }
    void combine(const Aggregator &o) { // ProjectAggregateQueue.scala:27-34
        switch (kind) {
        case CountAggr: counter += o.counter; break;
        case MaxDoubleAggr: if (o.dvalue > dvalue) dvalue = o.dvalue; break;
        case MinDoubleAggr: if (o.dvalue < dvalue) dvalue = o.dvalue; break;
        case MaxStringAggr: if (svalue.empty() || o.svalue > svalue) svalue = o.svalue; break;
        }
    }
    std::string repr() const {
        switch (kind) {
        case CountAggr: return std::to_string(counter);
        case MaxStringAggr: return svalue;
        default: return javaDoubleToString(dvalue);
        }
    }
};

using AggMap = std::vector<Aggregator>;                       // alias order = SELECT-list order (see Engine::executeAgg)
using AggMapTuple = std::pair<std::string, AggMap>;           // (groupKey, aggregators)

// ProjectAggOp(aggs, op, groupBy): one segment, fused on the GPU (scan+select kernel, LDS hash aggregation kernel)
class ProjectAggOp : public Operator<AggMapTuple> {
  public:
    ProjectAggOp(std::vector<Aggregator> aggs, std::shared_ptr<ColumnVectorOperator> op, std::vector<std::string> groupBy)
        : aggs_(std::move(aggs)), op_(std::move(op)), groupBy_(std::move(groupBy)) {}
    std::unique_ptr<Iterator<AggMapTuple>> iterator() override {
        std::vector<Leaf> leaves;
        std::shared_ptr<ScanOp> scan;
        if (auto sel = std::dynamic_pointer_cast<SelectOp>(op_)) scan = sel->chain(leaves);
        else scan = std::dynamic_pointer_cast<ScanOp>(op_);
        if (!scan) throw Exception("ProjectAggOp must sit on a ScanOp / SelectOp chain for the fused GPU path");
        SelectOp::checkConditions(leaves);
        const std::vector<Column> &cols = scan->cols();
        auto usedIndex = [&](const std::string &n) -> int32_t {
            for (size_t i = 0; i < cols.size(); ++i) if (cols[i].name == n) return (int32_t)i;
            throw Exception("NoSuchElementException: key not found: " + n);
        };
        // groupCols filters the BATCH columns by membership in groupBy (ProjectAggregate.scala:135-140)
        std::vector<int32_t> group;
        for (size_t i = 0; i < cols.size(); ++i)
            for (const auto &g : groupBy_) if (cols[i].name == g) { group.push_back((int32_t)i); break; }
        // aggsMap is keyed by alias: a later duplicate replaces an earlier one
        std::vector<Aggregator> aggs;
        for (const auto &a : aggs_) {
            bool replaced = false;
            for (auto &b : aggs) if (b.alias == a.alias) { b = a; replaced = true; }
            if (!replaced) aggs.push_back(a);
        }
        std::vector<imm3_aggregate> abi(aggs.size());
        for (size_t j = 0; j < aggs.size(); ++j) { abi[j].kind = aggs[j].abiKind(); abi[j].column = usedIndex(aggs[j].col); }
        QueryHandle h;
        scan->makeAggQuery(leaves, group, abi, h);
        auto it = std::make_unique<VectorIterator<AggMapTuple>>();
        it->items = decode(h, cols, group, aggs, abi);
        return it;
    }

    // the same aggregation as ONE table-level query (imm3_table): groups merged across segments on the GPU
    std::vector<AggMapTuple> runOn(imm3_table *table, const std::vector<Column> &cols, const std::vector<int32_t> &usedIdx, const std::vector<imm3_select> &sels) {
        std::vector<int32_t> group;
        for (size_t i = 0; i < cols.size(); ++i)
            for (const auto &g : groupBy_) if (cols[i].name == g) { group.push_back((int32_t)i); break; }
        std::vector<Aggregator> aggs;
        for (const auto &a : aggs_) {
            bool replaced = false;
            for (auto &b : aggs) if (b.alias == a.alias) { b = a; replaced = true; }
            if (!replaced) aggs.push_back(a);
        }
        std::vector<imm3_aggregate> abi(aggs.size());
        for (size_t j = 0; j < aggs.size(); ++j) {
            abi[j].kind = aggs[j].abiKind();
            abi[j].column = -1;
            for (size_t i = 0; i < cols.size(); ++i) if (cols[i].name == aggs[j].col) abi[j].column = (int32_t)i;
            if (abi[j].column < 0) throw Exception("NoSuchElementException: key not found: " + aggs[j].col);
        }
        auto scanOp = std::dynamic_pointer_cast<ScanOp>(op_);
        QueryHandle h;
        imm3Check(imm3_query_create_table_agg(scanOp->manager().ctx(), table, usedIdx.data(), (int32_t)usedIdx.size(), sels.data(), (int32_t)sels.size(),
                                              group.data(), (int32_t)group.size(), abi.data(), (int32_t)abi.size(), scanOp->table().blockSize, &h.q));
        return decode(h, cols, group, aggs, abi);
    }

  private:
    static std::vector<AggMapTuple> decode(QueryHandle &h, const std::vector<Column> &cols, const std::vector<int32_t> &group,
                                           const std::vector<Aggregator> &aggs, const std::vector<imm3_aggregate> &abi) {
        imm3Check(imm3_query_run(h.q));
        uint32_t n = 0;
        imm3Check(imm3_query_group_count(h.q, &n));
        std::vector<uint64_t> keys(n), counts(n);
        std::vector<uint32_t> first(n);
        std::vector<int64_t> vals((size_t)n * aggs.size());
        imm3Check(imm3_query_fetch_groups(h.q, keys.data(), first.data(), counts.data(), vals.data(), n));
        std::vector<AggMapTuple> items;
        for (uint32_t g = 0; g < n; ++g) {
            std::string key;
            int off = 0;
            for (size_t k = 0; k < group.size(); ++k) {
                const Column &c = cols[(size_t)group[k]];
                const int w = c.width();
                uint8_t raw[8] = {0};
                for (int b = 0; b < w; ++b) raw[b] = (uint8_t)(keys[g] >> (8 * (off + b)));
                ColumnVector v;
                v.type = c.columnType;
                v.width = w;
                v.data = raw;
                if (k) key += "_";
                key += v.value(0).toString(); // mkString("_"), ProjectAggregate.scala:144
                off += w;
            }
            AggMap m = aggs;
            for (size_t j = 0; j < aggs.size(); ++j) {
                const int64_t x = vals[(size_t)g * aggs.size() + j];
                switch (m[j].kind) {
                case Aggregator::CountAggr: m[j].counter = (long long)counts[g]; break;
                case Aggregator::MaxStringAggr: {
                    const int w = cols[(size_t)abi[j].column].width();
                    std::string sv((size_t)w, '\0');
                    for (int b = 0; b < w; ++b) sv[(size_t)b] = (char)((uint64_t)x >> (8 * (w - 1 - b)));
                    m[j].svalue = sv;
                    break;
                }
                default: m[j].dvalue = (double)x;
                }
            }
            items.emplace_back(key, std::move(m));
        }
        return items;
    }

    std::vector<Aggregator> aggs_;
    std::shared_ptr<ColumnVectorOperator> op_;
    std::vector<std::string> groupBy_;
};

// ---- Engine (Engine.scala:81-197) ----
class Engine {
  public:
    explicit Engine(GpuSegmentManager &sm) : sm_(sm) {}

    // Engine.getColumns (:85-106): (rec(query.select).toList ++ projectColumns).toSet.toList.  Scala's Set1..Set4 keep
    // insertion order, so for <= 4 distinct columns this is first-seen order (SURVEY A.1 rule 3); from the fifth distinct
    // column on, the set is an immutable.HashSet and the order is its hash trie's (scala_sets.hpp).
    static std::vector<Column> getColumns(const Query &q, const Table &table) {
        std::function<scalasets::ColumnSet(const SelectADT &)> rec = [&](const SelectADT &s) {
            scalasets::ColumnSet out;
            if (s.kind == SelectADT::And || s.kind == SelectADT::Or) { // rec(a) ++ rec(b): b's elements, in b's order, added to a
                out = rec(*s.op1);
                for (const auto &c : rec(*s.op2).toList()) out.add(c);
            } else if (s.kind == SelectADT::Select) out.add(table.getColumn(s.col));
            return out;
        };
        scalasets::ColumnSet all;
        for (const auto &c : rec(*q.select).toList()) all.add(c);
        if (q.project.kind == ProjectADT::Project) for (const auto &c : q.project.cols) all.add(table.getColumn(c));
        else if (q.project.kind == ProjectADT::ProjectAgg) {
            for (const auto &a : q.project.aggs) all.add(table.getColumn(a.col));
            for (const auto &g : q.project.groupBy) all.add(table.getColumn(g));
        }
        return all.toList();
    }
    // resolveSelectOps + runOps (:108-128, :237-245): left-to-right fold, AND/OR tag ignored
    static std::vector<Leaf> resolveSelectOps(const Query &q) {
        std::vector<Leaf> out;
        std::function<void(const SelectADT &)> rec = [&](const SelectADT &s) {
            if (s.kind == SelectADT::And || s.kind == SelectADT::Or) { rec(*s.op1); rec(*s.op2); }
            else if (s.kind == SelectADT::Select) out.push_back(Leaf{s.col, s.cond});
        };
        rec(*q.select);
        return out;
    }
    // ---- single-launch table path (imm3_table) ----
    struct TablePlan {
        imm3_table *table = nullptr;
        std::vector<Column> used;
        std::vector<int32_t> usedIdx;
        std::vector<imm3_select> sels;
        std::vector<std::string> blobs;
        std::vector<std::vector<int32_t>> lens;
    };
    // fills `p` and returns true when the whole table can run as one fused launch
    bool tablePlan(const Query &q, TablePlan &p) {
        const Table &table = sm_.sm.getTable(q.table);
        p.table = sm_.deviceTable(q.table);
        if (!p.table) return false;
        p.used = getColumns(q, table);
        for (const auto &c : p.used) p.usedIdx.push_back(table.columnIndex(c.name));
        const std::vector<Leaf> leaves = resolveSelectOps(q);
        SelectOp::checkConditions(leaves);
        p.sels.resize(leaves.size());
        p.blobs.resize(leaves.size());
        p.lens.resize(leaves.size());
        for (size_t i = 0; i < leaves.size(); ++i) {
            int32_t ci = -1;
            for (size_t k = 0; k < p.used.size(); ++k) if (p.used[k].name == leaves[i].col) { ci = (int32_t)k; break; }
            if (ci < 0) throw Exception("NoSuchElementException: next on empty iterator");
            const Column &c = p.used[(size_t)ci];
            if (leaves[i].cond.kind == SelectCondition::Match &&
                (c.columnType != ColumnType::STRING || c.width() != 2 || leaves[i].cond.values.empty() || leaves[i].cond.values.size() > 8))
                return false; // the tile kernels take 2-byte strings with <= 8 IN-list values
            p.sels[i] = imm3_select{};
            p.sels[i].column = ci;
            p.sels[i].cond = (int32_t)leaves[i].cond.kind;
            p.sels[i].value = leaves[i].cond.value;
            for (const auto &v : leaves[i].cond.values) { p.blobs[i] += v; p.lens[i].push_back((int32_t)v.size()); }
            p.sels[i].match_bytes = (const uint8_t *)p.blobs[i].data();
            p.sels[i].match_lens = p.lens[i].data();
            p.sels[i].n_match = (int32_t)p.lens[i].size();
        }
        return true;
    }

    // one fused pipeline per segment; rows in ascending segment order (the reference's order across segments is
    // unspecified: queue interleaving, Engine.scala:255); `limit` is global, as the consumer-side ProjectOp's is.
    // Engine.resolveProjectOp (:130-156): default aliases col_max / col_min / col_count; Min over a STRING column
    // becomes MaxStringAggr (the reference's own mapping, :145); Sum / Avg parse but are rejected (:152).
    static std::vector<Aggregator> resolveProjectOp(const ProjectADT &p, const Table &table) {
        std::vector<Aggregator> out;
        for (const auto &a : p.aggs) {
            const bool isStr = table.getColumn(a.col).columnType == ColumnType::STRING;
            switch (a.kind) {
            case Aggregate::Max: out.push_back(Aggregator::make(isStr ? Aggregator::MaxStringAggr : Aggregator::MaxDoubleAggr, a.col, a.alias.empty() ? a.col + "_max" : a.alias)); break;
            case Aggregate::Min: out.push_back(Aggregator::make(isStr ? Aggregator::MaxStringAggr : Aggregator::MinDoubleAggr, a.col, a.alias.empty() ? a.col + "_min" : a.alias)); break;
            case Aggregate::Count: out.push_back(Aggregator::make(Aggregator::CountAggr, a.col, a.alias.empty() ? a.col + "_count" : a.alias)); break;
            default: throw Exception("Unknown Aggregate type");
            }
        }
        return out;
    }
    // ProjectAgg: per-segment ProjectAggOp + ProjectAggregateQueueOp's combine by key (first arrival first; segments
    // ascending).  Rows list the aggregators' repr in SELECT-list order (the reference iterates a mutable.HashMap
    // keyed by alias, whose order is not reproduced).
    std::vector<AggMapTuple> executeAgg(const Query &q) {
        const Table &table = sm_.sm.getTable(q.table);
        const std::vector<Column> used = getColumns(q, table);
        const std::vector<Leaf> leaves = resolveSelectOps(q);
        const std::vector<Aggregator> aggs = resolveProjectOp(q.project, table);
        {   // one table-level aggregation query: groups come back already merged in (segment, row) first-seen order
            TablePlan p;
            if (tablePlan(q, p)) {
                auto scan = std::make_shared<ScanOp>(sm_, 0, q.table, p.used);
                ProjectAggOp op(aggs, scan, q.project.groupBy);
                return op.runOn(p.table, p.used, p.usedIdx, p.sels);
            }
        }
        auto mkScan = ScanOp::mkScanOp(sm_, q.table);
        std::vector<AggMapTuple> result;
        const int nseg = sm_.sm.getTableSegmentCount(table.name);
        for (int seg = 0; seg < nseg; ++seg) {
            std::shared_ptr<ColumnVectorOperator> op = mkScan(used, seg);
            for (const auto &l : leaves) op = SelectOp::mkSelectOp(l.col, l.cond)(op);
            ProjectAggOp agg(aggs, op, q.project.groupBy);
            auto it = agg.iterator();
            while (it->hasNext()) {
                AggMapTuple t = it->next();
                bool merged = false;
                for (auto &r : result)
                    if (r.first == t.first) {
                        for (size_t j = 0; j < r.second.size() && j < t.second.size(); ++j) r.second[j].combine(t.second[j]);
                        merged = true;
                        break;
                    }
                if (!merged) result.push_back(std::move(t));
            }
        }
        return result;
    }
    std::vector<Row> execute(const Query &q) {
        if (q.project.kind == ProjectADT::ProjectAgg) {
            std::vector<Row> rows;
            for (const auto &t : executeAgg(q)) {
                std::vector<Value> xs;
                for (const auto &a : t.second) xs.push_back(Value::ofString(a.repr()));
                rows.push_back(Row::fromSeq(std::move(xs)));
            }
            return rows;
        }
        if (q.project.kind != ProjectADT::Project) throw Exception("NoProject");
        {   // the whole table in ONE fused launch when it qualifies
            TablePlan p;
            if (tablePlan(q, p)) {
                std::vector<int32_t> proj;
                std::vector<Column> pcols;
                for (const auto &name : q.project.cols) {
                    bool found = false;
                    for (size_t k = 0; k < p.used.size() && !found; ++k)
                        if (p.used[k].name == name) { proj.push_back((int32_t)k); pcols.push_back(p.used[k]); found = true; }
                    if (!found) throw Exception("NoSuchElementException: key not found: " + name);
                }
                QueryHandle h;
                imm3Check(imm3_query_create_table(sm_.ctx(), p.table, p.usedIdx.data(), (int32_t)p.usedIdx.size(), p.sels.data(), (int32_t)p.sels.size(),
                                                  proj.data(), (int32_t)proj.size(), q.project.limit, sm_.sm.getTable(q.table).blockSize, &h.q));
                imm3Check(imm3_query_run(h.q));
                uint64_t n = 0;
                imm3Check(imm3_query_row_count(h.q, &n));
                std::vector<std::vector<uint8_t>> bufs(pcols.size());
                std::vector<void *> ptrs(pcols.size());
                for (size_t j = 0; j < pcols.size(); ++j) {
                    bufs[j].resize((size_t)std::max<uint64_t>(n, 1) * (size_t)pcols[j].width());
                    ptrs[j] = bufs[j].data();
                }
                imm3Check(imm3_query_fetch_rows(h.q, nullptr, ptrs.data(), n));
                std::vector<Row> rows;
                rows.reserve((size_t)n);
                for (uint64_t i = 0; i < n; ++i) {
                    std::vector<Value> xs;
                    for (size_t j = 0; j < pcols.size(); ++j) {
                        ColumnVector v;
                        v.type = pcols[j].columnType;
                        v.width = pcols[j].width();
                        v.data = bufs[j].data();
                        xs.push_back(v.value((int)i));
                    }
                    rows.push_back(Row::fromSeq(std::move(xs)));
                }
                return rows;
            }
        }
        const Table &table = sm_.sm.getTable(q.table);
        const std::vector<Column> used = getColumns(q, table);
        const std::vector<Leaf> leaves = resolveSelectOps(q);
        auto mkScan = ScanOp::mkScanOp(sm_, q.table);
        auto mkProj = ProjectOp::mkProjectOp(q.project.cols, q.project.limit);
        std::vector<Row> rows;
        const int nseg = sm_.sm.getTableSegmentCount(table.name);
        for (int seg = 0; seg < nseg; ++seg) {
            if (q.project.limit > 0 && (int)rows.size() >= q.project.limit) break;
            std::shared_ptr<ColumnVectorOperator> op = mkScan(used, seg);
            for (const auto &l : leaves) op = SelectOp::mkSelectOp(l.col, l.cond)(op);
            auto it = mkProj(op)->iterator();
            while (it->hasNext() && !(q.project.limit > 0 && (int)rows.size() >= q.project.limit)) rows.push_back(it->next());
        }
        return rows;
    }

  private:
    GpuSegmentManager &sm_;
};

} // namespace immutabledb
