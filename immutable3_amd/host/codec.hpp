// codec.hpp -- write side of the PFOR_INT block codec: PFORCodecInt.encode (core/src/main/scala/immutabledb/codec/
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

} // namespace codec
} // namespace immutabledb
