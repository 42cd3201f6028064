// codec.hpp -- write side of the compressed block codecs (PFOR_INT below, snappy further down).  PFOR_INT: PFORCodecInt.encode (core/src/main/scala/immutabledb/codec/
// PFORCodec.scala:19-31), which SegmentWriter.flush applies to every block of a PFOR_INT column
// (core/.../storage/Segment.scala:115-122).
//
//     val compressed = iic.compress(ints)                       // JavaFastPFOR 0.1.10 IntegratedIntCompressor
//     ByteBuffer.allocate(compressed.length * 4 + 8) ... putInt // big-endian words
//     bos.write(result.array())                                 // + the 8 spare bytes, all zero
//
// IntegratedIntCompressor = [n] ++ IntegratedBinaryPacking over the first n - n % 32 values ++ IntegratedVariableByte
// over the rest; one running "previous value" (0 at the start of the block) is shared by both.  The read side is the
// GPU (csrc/imm3_codec.hip); this header is host-only and also backs imm3_pfor_encode_block of the C ABI.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

namespace immutabledb {
namespace codec {

// Appends fixed-width fields to a little-endian bit stream of 32-bit words.
class BitWriter {
  public:
    explicit BitWriter(std::vector<uint32_t> &out) : out_(out) {}
    void put(uint32_t value, int width) {
        acc_ |= (uint64_t)value << fill_;
        fill_ += width;
        while (fill_ >= 32) {
            out_.push_back((uint32_t)acc_);
            acc_ >>= 32;
            fill_ -= 32;
        }
    }
    void flush() { // only called on a word boundary for whole mini-blocks (32 * width bits)
        if (fill_ > 0) out_.push_back((uint32_t)acc_);
        acc_ = 0;
        fill_ = 0;
    }

  private:
    std::vector<uint32_t> &out_;
    uint64_t acc_ = 0;
    int fill_ = 0;
};

inline int deltaWidth(uint32_t prev, const int32_t *v) { // Util.maxdiffbits over one mini-block of 32
    uint32_t any = 0;
    for (int i = 0; i < 32; ++i) {
        any |= (uint32_t)v[i] - prev;
        prev = (uint32_t)v[i];
    }
    return any == 0 ? 0 : 32 - __builtin_clz(any);
}

inline void packMini(uint32_t prev, const int32_t *v, int width, std::vector<uint32_t> &out) {
    if (width == 32) { // integratedpack32 copies the values, not the deltas
        for (int i = 0; i < 32; ++i) out.push_back((uint32_t)v[i]);
        return;
    }
    if (width == 0) return;
    BitWriter bw(out);
    for (int i = 0; i < 32; ++i) {
        bw.put((uint32_t)v[i] - prev, width);
        prev = (uint32_t)v[i];
    }
    bw.flush();
}

// The int[] IntegratedIntCompressor.compress returns for one block.
inline std::vector<uint32_t> compressInts(const int32_t *v, int32_t n) {
    std::vector<uint32_t> out;
    out.reserve((size_t)n + 16);
    out.push_back((uint32_t)n);
    uint32_t prev = 0;
    const int32_t minis = n / 32;
    int32_t m = 0;
    while (minis - m >= 4) { // four mini-blocks share one header word
        int w[4];
        uint32_t p = prev;
        for (int k = 0; k < 4; ++k) {
            w[k] = deltaWidth(p, v + 32 * (m + k));
            p = (uint32_t)v[32 * (m + k) + 31];
        }
        out.push_back(((uint32_t)w[0] << 24) | ((uint32_t)w[1] << 16) | ((uint32_t)w[2] << 8) | (uint32_t)w[3]);
        for (int k = 0; k < 4; ++k) {
            packMini(prev, v + 32 * (m + k), w[k], out);
            prev = (uint32_t)v[32 * (m + k) + 31];
        }
        m += 4;
    }
    for (; m < minis; ++m) { // the last one to three mini-blocks carry a header each
        const int w = deltaWidth(prev, v + 32 * m);
        out.push_back((uint32_t)w);
        packMini(prev, v + 32 * m, w, out);
        prev = (uint32_t)v[32 * m + 31];
    }
    if (n % 32) { // variable-byte deltas, stop bit on the last byte of each value, zero-padded to a word
        std::vector<uint8_t> bytes;
        for (int32_t i = 32 * minis; i < n; ++i) {
            uint32_t d = (uint32_t)v[i] - prev;
            prev = (uint32_t)v[i];
            for (; d > 127; d >>= 7) bytes.push_back((uint8_t)(d & 127));
            bytes.push_back((uint8_t)(d | 128));
        }
        bytes.resize((bytes.size() + 3) & ~(size_t)3, 0);
        for (size_t i = 0; i < bytes.size(); i += 4) {
            uint32_t w;
            std::memcpy(&w, bytes.data() + i, 4); // little-endian host
            out.push_back(w);
        }
    }
    return out;
}

// count word + one word per packed value + one header per mini-block + 5 bytes per variable-byte value, + the 8 spare bytes
inline size_t pforEncodeBound(int32_t n) { return ((size_t)n + (size_t)n / 32 + (size_t)(n % 32) / 4 + 8) * 4 + 8; }

// PFORCodecInt.encode(bytes): the block's little-endian int32 values -> the bytes SegmentWriter appends to the .dat
inline std::vector<uint8_t> pforEncodeBlock(const int32_t *v, int32_t n) {
    const std::vector<uint32_t> words = compressInts(v, n);
    std::vector<uint8_t> out(words.size() * 4 + 8, 0);
    for (size_t i = 0; i < words.size(); ++i) {
        const uint32_t w = words[i];
        out[4 * i] = (uint8_t)(w >> 24);
        out[4 * i + 1] = (uint8_t)(w >> 16);
        out[4 * i + 2] = (uint8_t)(w >> 8);
        out[4 * i + 3] = (uint8_t)w;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// Snappy-coded blocks: SnappyCodec.encode (core/src/main/scala/immutabledb/codec/SnappyCodec.scala:15-27) writes the
// block's raw value bytes through `new SnappyOutputStream(...)` of org.iq80.snappy 0.4:
//     "snappy\0" | per chunk of <= 32768 input bytes: flag (1 compressed / 0 stored), payload length (2 bytes, big-
//     endian), masked CRC-32C of the chunk's input bytes (4 bytes, big-endian), payload
// with the chunk stored compressed only when compressed / input <= 7/8.  The payload is raw Snappy (varint length,
// literal / copy elements).  Any valid Snappy stream is readable by any Snappy reader; the matcher below is a plain
// greedy one (one candidate per 4-byte hash, offsets below 64 KiB), not a byte-for-byte clone of iq80's.
// ---------------------------------------------------------------------------------------------------------------
inline uint32_t crc32c(const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 255u] ^ (c >> 8);
    return ~c;
}

inline uint32_t maskedCrc32c(const uint8_t *p, size_t n) {
    const uint32_t c = crc32c(p, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

class SnappyWriter { // raw Snappy elements appended to a byte vector
  public:
    explicit SnappyWriter(std::vector<uint8_t> &out) : out_(out) {}
    void varint(uint32_t v) {
        for (; v > 127; v >>= 7) out_.push_back((uint8_t)(v | 128));
        out_.push_back((uint8_t)v);
    }
    void literal(const uint8_t *p, size_t len) {
        if (!len) return;
        const size_t n = len - 1;
        if (n < 60) out_.push_back((uint8_t)(n << 2));
        else {
            const int bytes = n < 256 ? 1 : n < 65536 ? 2 : n < (1u << 24) ? 3 : 4;
            out_.push_back((uint8_t)((59 + bytes) << 2));
            for (int k = 0; k < bytes; ++k) out_.push_back((uint8_t)(n >> (8 * k)));
        }
        out_.insert(out_.end(), p, p + len);
    }
    void copy(size_t offset, size_t len) { // offset < 65536
        while (len) {
            size_t l = len < 64 ? len : 64;
            if (len > 64 && len - 64 < 4) l = 60; // never leave a remainder shorter than a match
            if (l >= 4 && l <= 11 && offset < 2048) {
                out_.push_back((uint8_t)(1u | ((l - 4) << 2) | ((offset >> 8) << 5)));
                out_.push_back((uint8_t)offset);
            } else {
                out_.push_back((uint8_t)(2u | ((l - 1) << 2)));
                out_.push_back((uint8_t)offset);
                out_.push_back((uint8_t)(offset >> 8));
            }
            len -= l;
        }
    }

  private:
    std::vector<uint8_t> &out_;
};

inline void snappyCompress(const uint8_t *in, size_t n, std::vector<uint8_t> &out) { // n <= 32768 here
    SnappyWriter w(out);
    w.varint((uint32_t)n);
    std::vector<int32_t> last(1u << 13, -1); // most recent position of each 4-byte hash
    size_t anchor = 0, i = 0;
    while (i + 4 <= n) {
        uint32_t four;
        std::memcpy(&four, in + i, 4);
        const uint32_t h = (four * 2654435761u) >> 19;
        const int32_t c = last[h];
        last[h] = (int32_t)i;
        if (c >= 0 && i - (size_t)c <= 65535 && std::memcmp(in + c, in + i, 4) == 0) {
            size_t len = 4;
            while (i + len < n && in[(size_t)c + len] == in[i + len]) ++len;
            w.literal(in + anchor, i - anchor);
            w.copy(i - (size_t)c, len);
            i += len;
            anchor = i;
        } else {
            ++i;
        }
    }
    w.literal(in + anchor, n - anchor);
}

inline size_t snappyEncodeBound(size_t n) { return 7 + (n / 32768 + 1) * 7 + n + n / 6 + 32; }

// SnappyCodec.encode(bytes): one storage block's raw value bytes -> the bytes SegmentWriter appends to the .dat
inline std::vector<uint8_t> snappyEncodeBlock(const uint8_t *in, size_t n) {
    std::vector<uint8_t> out = {'s', 'n', 'a', 'p', 'p', 'y', 0};
    std::vector<uint8_t> packed;
    for (size_t s = 0; s < n; s += 32768) {
        const size_t len = n - s < 32768 ? n - s : 32768;
        packed.clear();
        snappyCompress(in + s, len, packed);
        const bool compressed = (double)packed.size() / (double)len <= 7.0 / 8.0;
        const size_t plen = compressed ? packed.size() : len;
        const uint32_t crc = maskedCrc32c(in + s, len);
        const uint8_t hdr[7] = {(uint8_t)(compressed ? 1 : 0), (uint8_t)(plen >> 8), (uint8_t)plen,
                                (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
        out.insert(out.end(), hdr, hdr + 7);
        if (compressed) out.insert(out.end(), packed.begin(), packed.end());
        else out.insert(out.end(), in + s, in + s + len);
    }
    return out;
}

} // namespace codec
} // namespace immutabledb
