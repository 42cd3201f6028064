// scala_sets.hpp -- iteration order of a Scala 2.12 immutable Set[Column]: what decides Engine.getColumns' column order
// (engine/src/main/scala/immutabledb/engine/Engine.scala:105: (rec(query.select).toList ++ projectColumns).toSet.toList).
//
// The algorithm lives in scala-library 2.12.11 (build.sbt:2), a dependency that is not part of the reference checkout and
// cannot run here (no JVM); it is RESTATED from its published source (see immutable3_amd/scala_sets.py for the list of
// pieces).  PARITY UNPINNED at the reference boundary: pins are the murmur blocks against an independent implementation
// and Set(1 to 10).toList == List(5, 10, 1, 6, 9, 2, 7, 3, 8, 4), the library's well-known output (tests/test_host.py).
// With <= 4 distinct columns -- every query over the reference's own 3-column tables -- none of this is reached.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "schema.hpp"

namespace immutabledb {
namespace scalasets {

inline uint32_t javaStringHash(const std::string &s) { // ASCII / Latin-1 code units (column names, attribute keys)
    uint32_t h = 0;
    for (unsigned char ch : s) h = 31u * h + ch;
    return h;
}
inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline uint32_t mixLast(uint32_t h, uint32_t data) {
    uint32_t k = data * 0xcc9e2d51u;
    k = rotl(k, 15);
    k *= 0x1b873593u;
    return h ^ k;
}
inline uint32_t mix(uint32_t h, uint32_t data) {
    h = mixLast(h, data);
    h = rotl(h, 13);
    return h * 5u + 0xe6546b64u;
}
inline uint32_t finalizeHash(uint32_t h, uint32_t length) {
    h ^= length;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
inline uint32_t productHash(const std::vector<uint32_t> &fields) { // MurmurHash3.productHash(x, 0xcafebabe)
    uint32_t h = 0xcafebabeu;
    for (uint32_t f : fields) h = mix(h, f);
    return finalizeHash(h, (uint32_t)fields.size());
}
inline uint32_t mapHash(const std::vector<std::pair<std::string, std::string>> &entries) { // MurmurHash3.unorderedHash(tuples, "Map".hashCode)
    uint32_t a = 0, b = 0, n = 0, c = 1;
    for (const auto &kv : entries) {
        const uint32_t h = productHash({javaStringHash(kv.first), javaStringHash(kv.second)});
        a += h;
        b ^= h;
        if (h != 0) c *= h;
        ++n;
    }
    uint32_t h = javaStringHash("Map");
    h = mix(h, a);
    h = mix(h, b);
    h = mixLast(h, c);
    return finalizeHash(h, n);
}
inline uint32_t improve(uint32_t hcode) { // immutable.HashSet.improve
    uint32_t h = hcode + ~(hcode << 9);
    h ^= h >> 14;
    h += h << 4;
    return h ^ (h >> 10);
}
inline uint32_t columnHash(const Column &c) { // case class Column(name, columnType, codec, dtypeAttrs).hashCode
    return productHash({javaStringHash(c.name), (uint32_t)(int)c.columnType, (uint32_t)(int)c.codec, mapHash(c.dtypeAttrs)});
}
inline std::array<uint32_t, 7> trieKey(uint32_t hcode) { // HashTrieSet iteration: 5-bit chunks of the improved hash, low bits first
    const uint32_t h = improve(hcode);
    std::array<uint32_t, 7> k{};
    for (int i = 0; i < 7; ++i) k[(size_t)i] = (h >> (5 * i)) & 31u;
    return k;
}

// immutable.Set[Column] as far as its ITERATION ORDER goes
struct ColumnSet {
    std::vector<Column> small; // Set1..Set4: insertion order
    std::vector<Column> trie;  // HashSet (once a fifth distinct element arrives)
    bool hashed = false;
    void add(const Column &c) {
        auto &v = hashed ? trie : small;
        for (const auto &x : v)
            if (x == c) return;
        if (!hashed && small.size() == 4) { // Set4 + elem: new HashSet + (elem1, elem2, elem3, elem4, elem)
            trie = small;
            small.clear();
            hashed = true;
            trie.push_back(c);
            return;
        }
        v.push_back(c);
    }
    std::vector<Column> toList() const {
        if (!hashed) return small;
        std::vector<Column> out = trie;
        std::stable_sort(out.begin(), out.end(), [](const Column &x, const Column &y) { return trieKey(columnHash(x)) < trieKey(columnHash(y)); });
        return out;
    }
};

} // namespace scalasets
} // namespace immutabledb
