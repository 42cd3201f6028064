/*
 * imm3_oracle_snappy.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) for snappy-coded blocks.
 *
 * Reference: core/codec/SnappyCodec.scala:14-43.  SnappyCodec.encode writes the block's raw value bytes through
 * `new SnappyOutputStream(...)`; `decode` is `???` (SnappyCodec.scala:45), no CodecType names the codec
 * (Codec.scala:21-24) and nothing instantiates it -- so there is NO reference behaviour for reading such a block and
 * no way for a reference table to declare one.  What exists is the block FORMAT the encoder defines, restated here.
 *
 * THIRD-PARTY ALGORITHM: org.iq80.snappy:snappy:0.4 (project/Dependencies.scala), not under /root/reference and not
 * installable here.  Restated from the library's published source / the public Snappy format description:
 *   SnappyOutputStream (the pre-"framed" stream format of iq80 0.x):
 *     stream header   's' 'n' 'a' 'p' 'p' 'y' 0x00                                   (7 bytes, once per stream)
 *     per chunk of at most 32768 input bytes:
 *       flag          0x01 = snappy-compressed payload, 0x00 = stored
 *       length        payload bytes, 2 bytes big-endian
 *       checksum      masked CRC-32C of the chunk's UNCOMPRESSED bytes, 4 bytes big-endian;
 *                     mask(c) = ((c >>> 15) | (c << 17)) + 0xa282ead8
 *       payload       the chunk is stored compressed only if compressed / input <= 7/8
 *   raw Snappy payload: varint32 uncompressed length, then elements by tag & 3:
 *     00 literal   len-1 in tag>>2 (60..63 => 1..4 little-endian length bytes follow), then the bytes
 *     01 copy      len = 4 + ((tag>>2) & 7), offset = (tag>>5)<<8 | next byte
 *     10 copy      len = 1 + (tag>>2), offset = next 2 bytes little-endian
 *     11 copy      len = 1 + (tag>>2), offset = next 4 bytes little-endian
 *     (a copy may overlap its own output: bytes are produced one at a time)
 * PARITY UNPINNED at the reference boundary (nothing to run, nothing reads this format there).  The raw-Snappy layer
 * is pinned against an independent implementation available offline -- pyarrow's bundled Google snappy
 * (tests/test_oracle_snappy.py: its compressor's output must decode here, this file's compressor's output must decode
 * there).  The stream framing is pinned only by hand-made known-answer streams and the CRC-32C check value.
 */
#include "imm3_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- CRC-32C (Castagnoli, reflected 0x82F63B78) ---- */
uint32_t imm3o_crc32c(const uint8_t *p, int64_t n) {
    uint32_t c = 0xFFFFFFFFu;
    for (int64_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
    }
    return ~c;
}

uint32_t imm3o_crc32c_masked(const uint8_t *p, int64_t n) {
    const uint32_t c = imm3o_crc32c(p, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

/* ---- raw Snappy ---- */
int64_t imm3o_snappy_raw_uncompressed_length(const uint8_t *in, int64_t n) {
    uint32_t v = 0;
    for (int i = 0, shift = 0; i < 5 && i < n; i++, shift += 7) {
        v |= (uint32_t)(in[i] & 127) << shift;
        if (!(in[i] & 128)) return (int64_t)v;
    }
    return -1;
}

/* returns the number of bytes produced, -1 if the stream is malformed or does not fit `cap` */
int64_t imm3o_snappy_raw_decode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap) {
    int64_t ip = 0;
    uint32_t want = 0;
    int shift = 0;
    for (;;) {
        if (ip >= n || shift > 28) return -1;
        const uint8_t b = in[ip++];
        want |= (uint32_t)(b & 127) << shift;
        if (!(b & 128)) break;
        shift += 7;
    }
    if ((int64_t)want > cap) return -1;
    int64_t op = 0;
    while (ip < n) {
        const uint8_t tag = in[ip++];
        int64_t len, off = 0;
        switch (tag & 3) {
        case 0: {
            len = (tag >> 2) + 1;
            if (len > 60) {
                const int nb = (int)len - 60;
                if (ip + nb > n) return -1;
                uint32_t l = 0;
                for (int k = 0; k < nb; k++) l |= (uint32_t)in[ip + k] << (8 * k);
                ip += nb;
                len = (int64_t)l + 1;
            }
            if (ip + len > n || op + len > (int64_t)want) return -1;
            memcpy(out + op, in + ip, (size_t)len);
            ip += len;
            op += len;
            continue;
        }
        case 1:
            if (ip + 1 > n) return -1;
            len = 4 + ((tag >> 2) & 7);
            off = ((int64_t)(tag >> 5) << 8) | in[ip];
            ip += 1;
            break;
        case 2:
            if (ip + 2 > n) return -1;
            len = 1 + (tag >> 2);
            off = (int64_t)in[ip] | ((int64_t)in[ip + 1] << 8);
            ip += 2;
            break;
        default:
            if (ip + 4 > n) return -1;
            len = 1 + (tag >> 2);
            off = (int64_t)in[ip] | ((int64_t)in[ip + 1] << 8) | ((int64_t)in[ip + 2] << 16) | ((int64_t)in[ip + 3] << 24);
            ip += 4;
            break;
        }
        if (off == 0 || off > op || op + len > (int64_t)want) return -1;
        for (int64_t i = 0; i < len; i++) out[op + i] = out[op - off + i]; /* byte at a time: overlap repeats the pattern */
        op += len;
    }
    return op == (int64_t)want ? op : -1;
}

static int64_t emit_literal(uint8_t *out, int64_t op, const uint8_t *src, int64_t len) {
    const int64_t n = len - 1;
    if (n < 60) out[op++] = (uint8_t)(n << 2);
    else {
        int nb = n < (1 << 8) ? 1 : n < (1 << 16) ? 2 : n < (1 << 24) ? 3 : 4;
        out[op++] = (uint8_t)((59 + nb) << 2);
        for (int k = 0; k < nb; k++) out[op++] = (uint8_t)(n >> (8 * k));
    }
    memcpy(out + op, src, (size_t)len);
    return op + len;
}

static int64_t emit_copy(uint8_t *out, int64_t op, int64_t off, int64_t len) {
    while (len > 0) {
        int64_t l = len > 64 ? 64 : len;
        if (len > 64 && len < 68) l = 60; /* keep the remainder >= 4 */
        if (l >= 4 && l <= 11 && off < 2048) {
            out[op++] = (uint8_t)(1 | ((l - 4) << 2) | ((off >> 8) << 5));
            out[op++] = (uint8_t)off;
        } else {
            out[op++] = (uint8_t)(2 | ((l - 1) << 2));
            out[op++] = (uint8_t)off;
            out[op++] = (uint8_t)(off >> 8);
        }
        len -= l;
    }
    return op;
}

int64_t imm3o_snappy_raw_bound(int64_t n) { return 32 + n + n / 6; }

/* a plain greedy compressor (hash of 4 bytes, offsets < 65536): any valid Snappy stream will do for a reader */
int64_t imm3o_snappy_raw_encode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap) {
    if (cap < imm3o_snappy_raw_bound(n)) return -1;
    int64_t op = 0;
    uint32_t v = (uint32_t)n;
    while (v >= 128) { out[op++] = (uint8_t)(v | 128); v >>= 7; }
    out[op++] = (uint8_t)v;
    enum { HB = 12 };
    int32_t table[1 << HB];
    for (int i = 0; i < (1 << HB); i++) table[i] = -1;
    int64_t lit = 0, i = 0;
    while (i + 4 <= n) {
        uint32_t w;
        memcpy(&w, in + i, 4);
        const uint32_t h = (w * 0x1e35a7bdu) >> (32 - HB);
        const int64_t cand = table[h];
        table[h] = (int32_t)i;
        if (cand >= 0 && i - cand < 65536 && memcmp(in + cand, in + i, 4) == 0) {
            int64_t len = 4;
            while (i + len < n && in[cand + len] == in[i + len]) len++;
            if (i > lit) op = emit_literal(out, op, in + lit, i - lit);
            op = emit_copy(out, op, i - cand, len);
            i += len;
            lit = i;
        } else {
            i++;
        }
    }
    if (n > lit) op = emit_literal(out, op, in + lit, n - lit);
    return op;
}

/* ---- SnappyOutputStream framing of one storage block (SnappyCodec.encode) ---- */
static const uint8_t kHeader[7] = {'s', 'n', 'a', 'p', 'p', 'y', 0};

int64_t imm3o_snappy_block_bound(int64_t n) { return 7 + (n / 32768 + 1) * 7 + imm3o_snappy_raw_bound(n); }

int64_t imm3o_snappy_block_encode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap) {
    if (cap < imm3o_snappy_block_bound(n)) return -1;
    int64_t op = 0;
    memcpy(out, kHeader, 7);
    op = 7;
    uint8_t *tmp = (uint8_t *)malloc((size_t)imm3o_snappy_raw_bound(32768));
    if (!tmp) return -1;
    for (int64_t s = 0; s < n; s += 32768) {
        const int64_t len = n - s < 32768 ? n - s : 32768;
        const uint32_t crc = imm3o_crc32c_masked(in + s, len);
        const int64_t c = imm3o_snappy_raw_encode(in + s, len, tmp, imm3o_snappy_raw_bound(32768));
        const int compressed = c >= 0 && (double)c / (double)len <= 7.0 / 8.0;
        const int64_t plen = compressed ? c : len;
        out[op++] = compressed ? 1 : 0;
        out[op++] = (uint8_t)(plen >> 8);
        out[op++] = (uint8_t)plen;
        out[op++] = (uint8_t)(crc >> 24);
        out[op++] = (uint8_t)(crc >> 16);
        out[op++] = (uint8_t)(crc >> 8);
        out[op++] = (uint8_t)crc;
        memcpy(out + op, compressed ? tmp : in + s, (size_t)plen);
        op += plen;
    }
    free(tmp);
    return op;
}

/* total uncompressed bytes the block's chunks declare; -1 if malformed */
int64_t imm3o_snappy_block_length(const uint8_t *blk, int64_t n) {
    if (n < 7 || memcmp(blk, kHeader, 7) != 0) return -1;
    int64_t ip = 7, total = 0;
    while (ip < n) {
        if (ip + 7 > n) return -1;
        const int flag = blk[ip];
        const int64_t plen = ((int64_t)blk[ip + 1] << 8) | blk[ip + 2];
        ip += 7;
        if (flag > 1 || ip + plen > n) return -1;
        if (flag) {
            const int64_t u = imm3o_snappy_raw_uncompressed_length(blk + ip, plen);
            if (u < 0) return -1;
            total += u;
        } else {
            total += plen;
        }
        ip += plen;
    }
    return total;
}

/* SnappyInputStream(verifyChecksums = true) over the block: bytes produced, -1 if malformed / checksum mismatch */
int64_t imm3o_snappy_block_decode(const uint8_t *blk, int64_t n, uint8_t *out, int64_t cap) {
    if (n < 7 || memcmp(blk, kHeader, 7) != 0) return -1;
    int64_t ip = 7, op = 0;
    while (ip < n) {
        if (ip + 7 > n) return -1;
        const int flag = blk[ip];
        const int64_t plen = ((int64_t)blk[ip + 1] << 8) | blk[ip + 2];
        const uint32_t crc = ((uint32_t)blk[ip + 3] << 24) | ((uint32_t)blk[ip + 4] << 16) | ((uint32_t)blk[ip + 5] << 8) | blk[ip + 6];
        ip += 7;
        if (flag > 1 || ip + plen > n) return -1;
        int64_t got;
        if (flag) {
            got = imm3o_snappy_raw_decode(blk + ip, plen, out + op, cap - op);
            if (got < 0) return -1;
        } else {
            if (op + plen > cap) return -1;
            memcpy(out + op, blk + ip, (size_t)plen);
            got = plen;
        }
        if (imm3o_crc32c_masked(out + op, got) != crc) return -1;
        op += got;
        ip += plen;
    }
    return op;
}
