// storage.hpp -- the reference's on-disk format: reader (SegmentManager / Segment) and writer (SegmentWriter,
// LoaderCli's load loop).  Mirrors core/src/main/scala/immutabledb/storage/{Segment,SegmentManager}.scala and
// loader/src/main/scala/immutabledb/loader/LoaderCli.scala:113-154, including the loader quirk that a full
// segment holds segmentSize*blockSize + 1 rows in segmentSize + 1 blocks.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstring>
#include <functional>

#include "codec.hpp"
#include "schema.hpp"

namespace immutabledb {

// ---- DataType.stringToValue / valueToBytes (core/.../DataType.scala:31-71) ----
inline std::string stringToBytes(const Column &col, const std::string &s) {
    auto parseInt = [&](long long lo, long long hi) -> long long {
        size_t i = 0;
        if (i < s.size() && (s[i] == '+' || s[i] == '-')) ++i;
        if (i == s.size()) throw Exception("NumberFormatException: For input string: \"" + s + "\"");
        for (size_t k = i; k < s.size(); ++k)
            if (s[k] < '0' || s[k] > '9') throw Exception("NumberFormatException: For input string: \"" + s + "\"");
        if (s.size() - i > 12) throw Exception("NumberFormatException: For input string: \"" + s + "\"");
        const long long v = std::stoll(s);
        if (v < lo || v > hi) throw Exception("NumberFormatException: Value out of range. Value:\"" + s + "\"");
        return v;
    };
    switch (col.codec) {
    case CodecType::DENSE_INT:
    case CodecType::SNAPPY_INT:
    case CodecType::PFOR_INT: { // s.toInt, IntType.valueToBytes :40-47 (little-endian)
        const int32_t v = (int32_t)parseInt(INT32_MIN, INT32_MAX);
        std::string b(4, '\0');
        b[3] = (char)((v >> 24) & 0xFF);
        b[2] = (char)((v >> 16) & 0xFF);
        b[1] = (char)((v >> 8) & 0xFF);
        b[0] = (char)(v & 0xFF);
        return b;
    }
    case CodecType::SNAPPY_TINYINT:
    case CodecType::DENSE_TINYINT: // s.toByte
        return std::string(1, (char)(int8_t)parseInt(-128, 127));
    case CodecType::SNAPPY_STRING:
    case CodecType::DENSE_STRING: // value.getBytes(): NO padding / truncation to dtypeAttrs("size") (:69)
        return s;
    }
    throw Exception("");
}

// ---- SegmentMeta (Segment.scala:33-58; JSON key is the singular "blockOffset") ----
struct SegmentMeta {
    std::vector<int32_t> blockOffsets;
    static SegmentMeta load(const std::string &path) {
        SegmentMeta m;
        const json::Value j = json::parse(readFile(path)); // keep the document alive while iterating it
        for (const auto &v : j.at("blockOffset").arr) m.blockOffsets.push_back((int32_t)v.num);
        return m;
    }
    static void store(const std::string &path, const SegmentMeta &m) {
        json::Value v = json::Value::object();
        json::Value a = json::Value::array();
        for (int32_t o : m.blockOffsets) a.arr.push_back(json::Value::number(o));
        v.put("blockOffset", a);
        std::ofstream f(path, std::ios::binary | std::ios::trunc);
        f << json::dump(v);
    }
};

// ---- SegmentWriter (Segment.scala:70-152) ----
class SegmentWriter {
  public:
    SegmentWriter(int id, int blockSize, const std::string &tableName, const Column &column, const std::string &dataDir, int segmentSize)
        : id_(id), blockSize_(blockSize), tableName_(tableName), column_(column), dataDir_(dataDir), segmentSize_(segmentSize) {
        mkdirs(dataDir + "/" + tableName);
        datPath_ = dataDir + "/" + tableName + "/" + column.name + "_" + std::to_string(id) + ".dat";
        metaPath_ = dataDir + "/" + tableName + "/" + column.name + "_" + std::to_string(id) + ".meta";
        file_.open(datPath_, std::ios::binary | std::ios::trunc); // setLength(0)
        capacity_ = (size_t)blockSize * (size_t)column.width();   // ByteBuffer.allocateDirect(blockSize * dtype.size)
        blockBufferOffsets.push_back(0);
    }
    SegmentWriter newSegment() const { return SegmentWriter(id_ + 1, blockSize_, tableName_, column_, dataDir_, segmentSize_); }

    void write(const std::string &x) { // Segment.scala:99-112: the (blockSize+1)-th write flushes first
        if ((int)blockBufferOffsets.size() > segmentSize_) throw Exception("Segment full");
        if (recordsWritten_ >= blockSize_) flush();
        const std::string b = stringToBytes(column_, x);
        if (buf_.size() + b.size() > capacity_) throw Exception("BufferOverflowException");
        buf_ += b;
        ++recordsWritten_;
    }
    void flush() { // codec.encode(bytes) (Segment.scala:115-122): identity for DENSE_* (codec/DenseCodec.scala:18-22)
        if (column_.codec == CodecType::PFOR_INT) { // PFORCodecInt.encode (codec/PFORCodec.scala:19-31)
            std::vector<int32_t> vals(buf_.size() / 4);
            std::memcpy(vals.data(), buf_.data(), vals.size() * 4);
            const std::vector<uint8_t> enc = codec::pforEncodeBlock(vals.data(), (int32_t)vals.size());
            buf_.assign((const char *)enc.data(), enc.size());
        } else if (isSnappy(column_.codec)) { // SnappyCodec.encode (codec/SnappyCodec.scala:15-27)
            const std::vector<uint8_t> enc = codec::snappyEncodeBlock((const uint8_t *)buf_.data(), buf_.size());
            buf_.assign((const char *)enc.data(), enc.size());
        }
        file_.write(buf_.data(), (std::streamsize)buf_.size());
        blockBufferOffsets.push_back(blockBufferOffsets.back() + (int32_t)buf_.size());
        buf_.clear();
        recordsWritten_ = 0;
    }
    int remaining() const { return segmentSize_ - ((int)blockBufferOffsets.size() - 1); }
    void close() {
        if (!buf_.empty()) flush();
        SegmentMeta m;
        m.blockOffsets = blockBufferOffsets;
        SegmentMeta::store(metaPath_, m);
        file_.close();
    }
    std::vector<int32_t> blockBufferOffsets;

  private:
    int id_, blockSize_;
    std::string tableName_;
    Column column_;
    std::string dataDir_;
    int segmentSize_;
    std::string datPath_, metaPath_;
    std::ofstream file_;
    size_t capacity_ = 0;
    std::string buf_;
    int recordsWritten_ = 0;
};

inline std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && (unsigned char)s[a] <= ' ') ++a;
    while (b > a && (unsigned char)s[b - 1] <= ' ') --b;
    return s.substr(a, b - a);
}

inline std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    std::string cur;
    for (char c : s) {
        if (c == sep) { out.push_back(cur); cur.clear(); }
        else cur += c;
    }
    out.push_back(cur);
    // Java's String.split drops trailing empty strings
    while (out.size() > 1 && out.back().empty()) out.pop_back();
    return out;
}

// LoaderCli.main's load loop (LoaderCli.scala:113-154): the first CSV line is a header and is skipped, fields
// are split on ',' and trimmed, bound positionally to the table's columns.
inline void loadCsv(const std::string &dataDir, const Table &table, const std::string &csvPath, int segmentSize) {
    std::ifstream in(csvPath);
    if (!in) throw Exception(csvPath + " (No such file or directory)");
    std::string line;
    std::getline(in, line); // header
    TableIO::clear(dataDir, table);
    TableIO::store(dataDir, table);
    std::vector<SegmentWriter> segs;
    segs.reserve(table.columns.size());
    for (const auto &c : table.columns) segs.emplace_back(0, table.blockSize, table.name, c, dataDir, segmentSize);
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const std::vector<std::string> vals = split(line, ',');
        for (size_t idx = 0; idx < vals.size(); ++idx) {
            SegmentWriter &seg = segs.at(idx);
            if (seg.remaining() > 0) {
                seg.write(trim(vals[idx]));
            } else {
                seg.close();
                SegmentWriter next = seg.newSegment();
                segs[idx] = std::move(next);
                segs[idx].write(trim(vals[idx]));
            }
        }
    }
    for (auto &s : segs) s.close();
}

// ---- read side ----
struct MappedFile { // getByteBuffer (SegmentManager.scala:81-87): read-only mmap kept for the process lifetime
    const uint8_t *data = nullptr;
    size_t size = 0;
    MappedFile() = default;
    explicit MappedFile(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Exception(path + " (No such file or directory)");
        struct stat st {};
        ::fstat(fd, &st);
        size = (size_t)st.st_size;
        if (size) {
            void *p = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) { ::close(fd); throw Exception("mmap failed: " + path); }
            data = (const uint8_t *)p;
        }
        ::close(fd);
    }
};

// Segment (Segment.scala:154-181): buffer + block offsets; the iterator yields one block per next() by relative
// gets from a rewound buffer, i.e. from a running cursor.
class Segment {
  public:
    Segment(int id, const MappedFile *buf, const SegmentMeta *meta) : id_(id), buf_(buf), meta_(meta) {}
    int id() const { return id_; }
    struct Block { const uint8_t *data; size_t size; };
    class BlockIterator {
      public:
        explicit BlockIterator(const Segment *s) : s_(s) {}
        bool hasNext() const { return position_ + 1 < (int)s_->meta_->blockOffsets.size(); }
        Block next() {
            const auto &o = s_->meta_->blockOffsets;
            const int64_t len = (int64_t)o[(size_t)position_ + 1] - (int64_t)o[(size_t)position_];
            if (len < 0) throw Exception("NegativeArraySizeException");
            if (cursor_ + (size_t)len > s_->buf_->size) throw Exception("BufferUnderflowException");
            Block b{s_->buf_->data + cursor_, (size_t)len};
            cursor_ += (size_t)len;
            ++position_;
            return b;
        }
      private:
        const Segment *s_;
        int position_ = 0;
        size_t cursor_ = 0;
    };
    BlockIterator iterator() const { return BlockIterator(this); }
    const MappedFile *buffer() const { return buf_; }
    const SegmentMeta *meta() const { return meta_; }

  private:
    int id_;
    const MappedFile *buf_;
    const SegmentMeta *meta_;
};

// SegmentManager (SegmentManager.scala:20-111).  Segment order = LEXICOGRAPHIC filename order (:38-42, :61-65).
class SegmentManager {
  public:
    explicit SegmentManager(const std::string &dataDir) : dataDir_(dataDir) {
        std::vector<std::string> dirs = listDir(dataDir, true);
        std::sort(dirs.begin(), dirs.end());
        for (const auto &d : dirs) tables.push_back(TableIO::load(dataDir, d));
        for (const auto &t : tables) {
            std::vector<std::string> files = listDir(dataDir + "/" + t.name, false);
            std::sort(files.begin(), files.end());
            for (const auto &c : t.columns) {
                const std::string key = t.name + "." + c.name;
                const std::string prefix = c.name + "_";
                for (const auto &f : files) {
                    if (f.compare(0, prefix.size(), prefix) != 0) continue;
                    const std::string path = dataDir + "/" + t.name + "/" + f;
                    if (f.size() > 4 && f.compare(f.size() - 4, 4, ".dat") == 0) segments[key].emplace_back(path);
                    else if (f.size() > 5 && f.compare(f.size() - 5, 5, ".meta") == 0) segmentsMeta[key].push_back(SegmentMeta::load(path));
                }
                segments[key];
                segmentsMeta[key];
            }
        }
    }
    const Table &getTable(const std::string &tableName) const {
        for (const auto &t : tables)
            if (t.name == tableName) return t;
        throw Exception("Table " + tableName + " does not exist in SegmentManager");
    }
    int getTableSegmentCount(const std::string &tableName) const {
        const Table &t = getTable(tableName);
        return (int)segments.at(tableName + "." + t.columns.front().name).size();
    }
    Segment getSegment(int id, const std::string &tableName, const std::string &columnName) const {
        const std::string key = tableName + "." + columnName;
        return Segment(id, &segments.at(key).at((size_t)id), &segmentsMeta.at(key).at((size_t)id));
    }
    std::vector<Segment> getSegments(const std::string &tableName, const std::string &columnName) const {
        std::vector<Segment> out;
        const std::string key = tableName + "." + columnName;
        for (size_t i = 0; i < segments.at(key).size(); ++i) out.push_back(getSegment((int)i, tableName, columnName));
        return out;
    }
    std::vector<Table> tables;
    std::map<std::string, std::vector<MappedFile>> segments;
    std::map<std::string, std::vector<SegmentMeta>> segmentsMeta;

  private:
    std::string dataDir_;
};

} // namespace immutabledb
