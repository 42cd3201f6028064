/*
 * imm3_oracle_pfor.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the PFOR_INT block codec.
 *
 * What a PFOR_INT block IS is defined by the reference's ENCODER, core/codec/PFORCodec.scala:19-31:
 *     val compressed = iic.compress(<the block's ints>)          // iic = new IntegratedIntCompressor()
 *     ByteBuffer.allocate(compressed.length * 4 + 8); putInt each  // java.nio default order: BIG-endian
 *     bos.write(result.array())                                  // the whole backing array: + 8 zero bytes
 * The reference's own DECODER is broken (PFORCodec.scala:43-50 reads nothing and hands an empty array to
 * iic.uncompress, which throws), so there is no reference decode behaviour to match: this file restates the decode
 * the encoder implies (IntegratedIntCompressor.uncompress of the big-endian ints).
 *
 * THIRD-PARTY ALGORITHM: me.lemire.integercompression:JavaFastPFOR:0.1.10 (project/Dependencies.scala:4), not under
 * /root/reference and not installable here.  Restated from the library's published algorithm:
 *   IntegratedIntCompressor()            = SkippableIntegratedComposition(IntegratedBinaryPacking, IntegratedVariableByte)
 *     compress(in):   out[0] = in.length; headlessCompress(in, ..., out from 1, initvalue = 0)
 *   IntegratedBinaryPacking (delta + bit packing, mini-blocks of 32 values):
 *     takes floor(n / 32) * 32 values.  While >= 4 mini-blocks remain: one header word
 *     (b1<<24)|(b2<<16)|(b3<<8)|b4, then b1, b2, b3, b4 packed words.  Each leftover mini-block: one header word = b,
 *     then b packed words.  b = bits(OR of the 32 wrapping deltas, the first against the running init value);
 *     value i occupies bits [i*b, (i+1)*b) of the mini-block's little-endian bit stream; b == 32 stores the 32 VALUES
 *     themselves (no delta); b == 0 stores nothing.  The init value becomes the mini-block's last value.
 *   IntegratedVariableByte (the n % 32 trailing values): unsigned delta in 7-bit groups, low group first, the LAST
 *     byte of a value carries 0x80; bytes are packed little-endian into words, zero-padded to a whole word.
 * PARITY UNPINNED: no golden vector of that library is available offline; the restatement is pinned only by
 * hand-derived known-answer blocks (tests/test_oracle_pfor.py), an independent numpy restatement (oracle_np.py)
 * and encode -> decode round trips.
 */
#include "imm3_oracle.h"

#include <stdlib.h>
#include <string.h>

static uint32_t bits_of(uint32_t mask) { /* Util.bits: 32 - numberOfLeadingZeros */
    uint32_t b = 0;
    while (mask) { b++; mask >>= 1; }
    return b;
}

static uint32_t maxdiffbits(int32_t init, const int32_t *v, int n) { /* Util.maxdiffbits */
    uint32_t mask = (uint32_t)v[0] - (uint32_t)init;
    for (int k = 1; k < n; k++) mask |= (uint32_t)v[k] - (uint32_t)v[k - 1];
    return bits_of(mask);
}

/* IntegratedBitPacking.integratedpack: 32 values -> b words; returns words written */
static int pack32(int32_t init, const int32_t *v, uint32_t *out, uint32_t b) {
    if (b == 0) return 0;
    if (b == 32) { /* integratedpack32: System.arraycopy of the values */
        for (int i = 0; i < 32; i++) out[i] = (uint32_t)v[i];
        return 32;
    }
    memset(out, 0, b * sizeof(uint32_t));
    uint32_t prev = (uint32_t)init;
    for (int i = 0; i < 32; i++) {
        const uint32_t d = (uint32_t)v[i] - prev;
        prev = (uint32_t)v[i];
        const uint32_t bit = (uint32_t)i * b, w = bit >> 5, s = bit & 31;
        out[w] |= d << s;
        if (s + b > 32) out[w + 1] |= d >> (32 - s);
    }
    return (int)b;
}

static void unpack32(int32_t init, const uint32_t *in, int32_t *v, uint32_t b) {
    if (b == 32) {
        for (int i = 0; i < 32; i++) v[i] = (int32_t)in[i];
        return;
    }
    const uint32_t mask = b == 0 ? 0u : ((1u << b) - 1u);
    uint32_t prev = (uint32_t)init;
    for (int i = 0; i < 32; i++) {
        uint32_t d = 0;
        if (b) {
            const uint32_t bit = (uint32_t)i * b, w = bit >> 5, s = bit & 31;
            d = in[w] >> s;
            if (s + b > 32) d |= in[w + 1] << (32 - s);
            d &= mask;
        }
        prev += d;
        v[i] = (int32_t)prev;
    }
}

static void put_be(uint8_t *p, uint32_t w) {
    p[0] = (uint8_t)(w >> 24); p[1] = (uint8_t)(w >> 16); p[2] = (uint8_t)(w >> 8); p[3] = (uint8_t)w;
}
static uint32_t get_be(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

int64_t imm3o_pfor_encode_bound(int32_t n) { return ((int64_t)n + 1024 + 1) * 4 + 8; }

/* PFORCodecInt.encode (PFORCodec.scala:19-31) of one block of n values.  Returns bytes written, -1 if cap is too small. */
int64_t imm3o_pfor_encode_block(const int32_t *vals, int32_t n, uint8_t *out, int64_t cap) {
    uint32_t *w = (uint32_t *)malloc(((size_t)n + 1024 + 64) * sizeof(uint32_t)); /* compress(): new int[input.length + 1024] */
    if (!w) return -1;
    size_t pos = 0;
    w[pos++] = (uint32_t)n;
    int32_t init = 0;
    const int packed = n / 32 * 32;
    int s = 0;
    for (; s + 128 <= packed; s += 128) { /* groups of four mini-blocks */
        uint32_t b[4];
        int32_t in = init;
        for (int k = 0; k < 4; k++) {
            b[k] = maxdiffbits(in, vals + s + 32 * k, 32);
            in = vals[s + 32 * k + 31];
        }
        w[pos++] = (b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3];
        for (int k = 0; k < 4; k++) {
            pos += (size_t)pack32(init, vals + s + 32 * k, w + pos, b[k]);
            init = vals[s + 32 * k + 31];
        }
    }
    for (; s < packed; s += 32) { /* leftover mini-blocks, one header each */
        const uint32_t b = maxdiffbits(init, vals + s, 32);
        w[pos++] = b;
        pos += (size_t)pack32(init, vals + s, w + pos, b);
        init = vals[s + 31];
    }
    if (n > packed) { /* IntegratedVariableByte over the tail */
        uint8_t bytes[32 * 5 + 4];
        size_t nb = 0;
        for (int k = packed; k < n; k++) {
            uint32_t d = (uint32_t)vals[k] - (uint32_t)init;
            init = vals[k];
            while (d >= 128) { bytes[nb++] = (uint8_t)(d & 127); d >>= 7; }
            bytes[nb++] = (uint8_t)(d | 128);
        }
        while (nb % 4) bytes[nb++] = 0;
        for (size_t i = 0; i < nb; i += 4)
            w[pos++] = (uint32_t)bytes[i] | ((uint32_t)bytes[i + 1] << 8) | ((uint32_t)bytes[i + 2] << 16) | ((uint32_t)bytes[i + 3] << 24);
    }
    const int64_t need = (int64_t)pos * 4 + 8;
    if (need > cap) { free(w); return -1; }
    for (size_t i = 0; i < pos; i++) put_be(out + 4 * i, w[i]);
    memset(out + 4 * pos, 0, 8);
    free(w);
    return need;
}

/* number of values the block declares (its first big-endian word); -1 if the block is too short */
int32_t imm3o_pfor_block_count(const uint8_t *blk, int64_t len) {
    if (len < 4) return -1;
    return (int32_t)get_be(blk);
}

/* IntegratedIntCompressor.uncompress of the block's big-endian words.  Returns the number of values, or
 * -1 for a malformed block (short, a width above 32, data running past the block). */
int32_t imm3o_pfor_decode_block(const uint8_t *blk, int64_t len, int32_t *out, int32_t cap) {
    if (len < 4 || len % 4) return -1;
    const int64_t nw = len / 4;
    const int32_t n = (int32_t)get_be(blk);
    if (n < 0 || n > cap) return -1;
    uint32_t *w = (uint32_t *)malloc(((size_t)nw + 2) * sizeof(uint32_t));
    if (!w) return -1;
    for (int64_t i = 0; i < nw; i++) w[i] = get_be(blk + 4 * i);
    w[nw] = w[nw + 1] = 0;
    int64_t pos = 1;
    int32_t init = 0;
    const int packed = n / 32 * 32;
    int s = 0;
    int bad = 0;
    for (; s + 128 <= packed && !bad; s += 128) {
        if (pos >= nw) { bad = 1; break; }
        const uint32_t h = w[pos++];
        for (int k = 0; k < 4; k++) {
            const uint32_t b = (h >> (24 - 8 * k)) & 255u;
            if (b > 32 || pos + b > nw) { bad = 1; break; }
            unpack32(init, w + pos, out + s + 32 * k, b);
            pos += b;
            init = out[s + 32 * k + 31];
        }
    }
    for (; s < packed && !bad; s += 32) {
        if (pos >= nw) { bad = 1; break; }
        const uint32_t b = w[pos++];
        if (b > 32 || pos + b > nw) { bad = 1; break; }
        unpack32(init, w + pos, out + s, b);
        pos += b;
        init = out[s + 31];
    }
    if (!bad && n > packed) {
        int64_t byte = pos * 4;
        for (int k = packed; k < n; k++) {
            uint32_t v = 0;
            int shift = 0;
            for (;;) {
                if (byte >= nw * 4 || shift > 28) { bad = 1; break; }
                const uint32_t c = (w[byte >> 2] >> (8 * (byte & 3))) & 255u;
                byte++;
                v += (c & 127u) << shift;
                if (c & 128u) break;
                shift += 7;
            }
            if (bad) break;
            init = (int32_t)((uint32_t)init + v);
            out[k] = init;
        }
    }
    free(w);
    return bad ? -1 : n;
}
