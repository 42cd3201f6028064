// imm3_snappy.hip -- snappy-coded blocks (gfx950, wave64): decode to the dense column.
//
// Reference: core/src/main/scala/immutabledb/codec/SnappyCodec.scala:14-43.  SnappyCodec.encode writes a block's raw
// value bytes through iq80 snappy 0.4's SnappyOutputStream:
//     "snappy\0"                                            stream header, once per block (one stream per encode call)
//     per chunk of <= 32768 input bytes:  flag (1 = raw-Snappy payload, 0 = stored) | payload length, 2 bytes
//                                         big-endian | masked CRC-32C of the uncompressed chunk, 4 bytes big-endian | payload
//     raw Snappy payload: varint uncompressed length, then literal / copy elements (tag & 3)
// The reference cannot read the format back (`decode = ???`, SnappyCodec.scala:45), has no CodecType for it and never
// instantiates the codec; IMM3_SNAPPY_* column codecs are therefore an extension of this library (include/imm3.h).
//
// LZ decoding is serial per chunk -- every element's position depends on the previous one -- so it is not fused into
// the filter: k_snappy_decode expands a segment's blocks ONCE into the dense column kept in HBM (as k_pfor_decode
// does), and every query then runs the dense kernels at HBM rate.  One wave per storage block:
//   * the block's bytes are staged in LDS (aligned dword loads; the block may start at any byte);
//   * elements are parsed wave-uniformly (one LDS read of the next 8 bytes, fields taken with v_readlane), and each
//     literal / copy is executed by all 64 lanes (an overlapping copy repeats its first `offset` bytes: lane i takes
//     byte i mod offset of the pattern);
//   * the chunk is assembled in LDS (copies reference up to 32 KiB back), its CRC-32C is verified (per-lane slices
//     combined with GF(2) shifts), and it is written to the column with coalesced stores.
#include "imm3_internal.h"
#include "imm3_device.h"
#include <hip/hip_ext.h>

namespace imm3 {

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ---------------------------------------------------------------------------------------------
// k_snappy_sizes: uncompressed bytes each block declares (sum over its chunks); 0xFFFFFFFF if malformed.
// One thread per block (a block of the default 1024 rows is a single chunk).
// ---------------------------------------------------------------------------------------------
__global__ void k_snappy_sizes(const uint8_t *data, const uint32_t *block_off, int64_t n_blocks, uint32_t *sizes, uint32_t *max_chunk) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_blocks) return;
    const uint8_t *p = data + block_off[k];
    const uint32_t n = block_off[k + 1] - block_off[k];
    uint32_t total = 0, biggest = 0;
    bool bad = n < 7;
    if (!bad) {
        const uint8_t hdr[7] = {'s', 'n', 'a', 'p', 'p', 'y', 0};
        for (int i = 0; i < 7; ++i) bad |= p[i] != hdr[i];
    }
    uint32_t ip = 7;
    while (!bad && ip < n) {
        if (ip + 7 > n) { bad = true; break; }
        const uint32_t flag = p[ip], plen = ((uint32_t)p[ip + 1] << 8) | p[ip + 2];
        ip += 7;
        if (flag > 1 || ip + plen > n) { bad = true; break; }
        uint32_t u = plen;
        if (flag) {
            u = 0;
            bool done = false;
            for (uint32_t i = 0, shift = 0; i < 5 && i < plen; ++i, shift += 7) {
                u |= (uint32_t)(p[ip + i] & 127u) << shift;
                if (!(p[ip + i] & 128u)) { done = true; break; }
            }
            bad |= !done;
        }
        bad |= u > 32768u; // SnappyOutputStream never puts more than 32768 input bytes into a chunk
        total += u;
        biggest = biggest > u ? biggest : u;
        ip += plen;
    }
    sizes[k] = bad ? 0xFFFFFFFFu : total;
    if (!bad) atomicMax(max_chunk, biggest);
}

// ---------------------------------------------------------------------------------------------
// CRC-32C (reflected, polynomial 0x82F63B78) pieces
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t crc_byte(const uint32_t *tab, uint32_t c, uint32_t b) { return tab[(c ^ b) & 255u] ^ (c >> 8); }

// a * b in GF(2)[x] / P, reflected bit order (bit 31 = x^0): used to move a slice's CRC past the bytes that follow it
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
        p ^= b & (0u - (a >> 31));
        a <<= 1;
        b = (b >> 1) ^ (0x82F63B78u & (0u - (b & 1u)));
    }
    return p;
}

// ---------------------------------------------------------------------------------------------
// k_snappy_decode: block k -> out bytes [row_base[k] * width, ...).  One wave per work-group; dynamic LDS =
// in_cap (staged block, + 8 slack) + out_cap (one chunk) + the 1 KiB CRC table.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_snappy_decode(const SnappyArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    uint32_t *crc_tab = (uint32_t *)s_mem;
    uint8_t *in = s_mem + 1024;
    uint8_t *outb = in + a.in_cap;
    const int lane = threadIdx.x;
    // CRC table: 4 entries per lane
    for (int e = lane; e < 256; e += 64) {
        uint32_t c = (uint32_t)e;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
        crc_tab[e] = c;
    }
    bool any_bad = false;
    for (int64_t k = blockIdx.x; k < a.n_blocks; k += gridDim.x) {
        const uint32_t o = a.block_off[k], e = a.block_off[k + 1];
        const uint32_t blen = e - o;
        const uint32_t skew = o & 3u; // the block may start on any byte: stage from the aligned dword below it
        const uint32_t expect = (a.row_base[k + 1] - a.row_base[k]) * (uint32_t)a.width;
        uint8_t *dst = a.out + (uint64_t)a.row_base[k] * (uint64_t)a.width;
        bool bad = blen + skew + 8 > (uint32_t)a.in_cap || blen < 7;
        if (!bad) {
            const uint32_t *src = (const uint32_t *)(a.data + (o - skew));
            const uint32_t nd = (blen + skew + 3) >> 2;
            // eight loads in flight per lane: one load per trip would be one HBM round trip per 256 bytes of the block
            for (uint32_t base = 0; base < nd; base += 512) {
                uint32_t r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t i = base + 64u * u + (uint32_t)lane;
                    r[u] = __builtin_nontemporal_load(src + (i < nd ? i : nd - 1));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t i = base + 64u * u + (uint32_t)lane;
                    if (i < nd) ((uint32_t *)in)[i] = r[u];
                }
            }
        }
        lds_wave_sync();
        const uint8_t *ib = in + skew;
        if (!bad) bad = !(ib[0] == 's' && ib[1] == 'n' && ib[2] == 'a' && ib[3] == 'p' && ib[4] == 'p' && ib[5] == 'y' && ib[6] == 0);
        bad = uni(bad) != 0;
        uint32_t ip = 7, done = 0;
        while (!bad && ip < blen) {
            if (ip + 7 > blen) { bad = true; break; }
            const uint32_t hb = lane < 7 ? ib[ip + lane] : 0u; // chunk header: one LDS read, fields by readlane
            const uint32_t flag = (uint32_t)__builtin_amdgcn_readlane((int)hb, 0);
            const uint32_t plen = ((uint32_t)__builtin_amdgcn_readlane((int)hb, 1) << 8) | (uint32_t)__builtin_amdgcn_readlane((int)hb, 2);
            const uint32_t crc_want = ((uint32_t)__builtin_amdgcn_readlane((int)hb, 3) << 24) | ((uint32_t)__builtin_amdgcn_readlane((int)hb, 4) << 16) |
                                      ((uint32_t)__builtin_amdgcn_readlane((int)hb, 5) << 8) | (uint32_t)__builtin_amdgcn_readlane((int)hb, 6);
            ip += 7;
            if (flag > 1 || ip + plen > blen) { bad = true; break; }
            uint32_t ulen = plen;
            if (flag == 0) { // stored chunk
                if (plen > (uint32_t)a.out_cap) { bad = true; break; }
                for (uint32_t i = lane; i < plen; i += 64) outb[i] = ib[ip + i];
            } else {
                // varint preamble
                const uint32_t end = ip + plen;
                uint32_t want = 0, pos = ip;
                {
                    const uint32_t vb = lane < 5 && ip + lane < end ? ib[ip + lane] : 0u;
                    bool ok = false;
                    for (int i = 0; i < 5 && !ok; ++i) {
                        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)vb, i);
                        want |= (b & 127u) << (7 * i);
                        ++pos;
                        ok = !(b & 128u);
                    }
                    if (!ok || pos > end || want > (uint32_t)a.out_cap) { bad = true; break; }
                }
                uint32_t op = 0;
                // Elements are parsed 64 byte positions at a time: lane l decodes the tag that WOULD start at pos + l (kind,
                // length, offset, header size) in vector code; the serial walk along the real element starts then costs two
                // v_readlane and a handful of scalar instructions per element instead of a full scalar decode (the scalar unit
                // is shared by the CU's four SIMDs and was the bottleneck: 150 scalar instructions per element).
                while (pos < end) {
                    const uint32_t p = pos + (uint32_t)lane;
                    const uint32_t tag = p < end ? ib[p] : 0u;
                    const uint32_t c1 = p + 1 < end ? ib[p + 1] : 0u, c2 = p + 2 < end ? ib[p + 2] : 0u;
                    const uint32_t c3 = p + 3 < end ? ib[p + 3] : 0u, c4 = p + 4 < end ? ib[p + 4] : 0u;
                    const uint32_t le = c1 | (c2 << 8) | (c3 << 16) | (c4 << 24);
                    const uint32_t kind = tag & 3u;
                    uint32_t len = (tag >> 2) + 1u, off = 0u, hdr = 1u;
                    if (kind == 0) {
                        if (len > 60u) {
                            const uint32_t nb = len - 60u;
                            len = (nb == 4u ? le : (le & ((1u << (8u * nb)) - 1u))) + 1u;
                            hdr = 1u + nb;
                        }
                    } else if (kind == 1) { len = 4u + ((tag >> 2) & 7u); off = ((tag >> 5) << 8) | c1; hdr = 2u; }
                    else if (kind == 2) { off = le & 0xFFFFu; hdr = 3u; }
                    else { off = le; hdr = 5u; }
                    if (len > 0x00FFFFFFu) len = 0x00FFFFFFu; // cannot be valid (a chunk holds <= 32768 bytes): fails the bound checks
                    const uint32_t packed = len | (hdr << 24) | (kind << 28);
                    uint32_t s = 0;
                    while (s < 64u && pos + s < end) {
                        const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)packed, (int)s);
                        const uint32_t eoff = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)s);
                        const uint32_t elen = pk & 0x00FFFFFFu, ehdr = (pk >> 24) & 15u, ekind = pk >> 28;
                        const uint32_t src = pos + s + ehdr;
                        if (ekind == 0) {
                            if (src + elen > end || op + elen > want) { bad = true; break; }
                            for (uint32_t i = lane; i < elen; i += 64) outb[op + i] = ib[src + i];
                            s += ehdr + elen;
                        } else {
                            if (src > end || eoff == 0 || eoff > op || op + elen > want) { bad = true; break; }
                            if ((uint32_t)lane < elen) { // elen <= 64: one byte per lane; an overlapping copy repeats its first eoff bytes
                                uint32_t r = (uint32_t)lane;
                                if (eoff < elen) { // lane mod eoff without an integer division (both < 64)
                                    const uint32_t q = (uint32_t)((float)lane * __frcp_rn((float)eoff));
                                    r = (uint32_t)lane - q * eoff;
                                    r = (int32_t)r < 0 ? r + eoff : (r >= eoff ? r - eoff : r);
                                }
                                outb[op + lane] = outb[op - eoff + r];
                            }
                            s += ehdr;
                        }
                        op += elen;
                    }
                    if (bad) break;
                    pos += s;
                }
                if (bad) break;
                if (op != want) { bad = true; break; }
                ulen = want;
            }
            lds_wave_sync();
            // CRC-32C of the chunk: lane l folds bytes [l * per, (l + 1) * per), then the slice CRCs are moved past the
            // bytes behind them (multiplication by x^(8 * bytes) mod P) and XORed together
            {
                const uint32_t per = (ulen + 63) >> 6;
                const uint32_t lo = (uint32_t)lane * per, hi = lo + per < ulen ? lo + per : ulen;
                uint32_t c = lane == 0 ? 0xFFFFFFFFu : 0u;
                for (uint32_t i = lo; i < hi; ++i) c = crc_byte(crc_tab, c, outb[i]);
                const uint32_t after = hi < ulen ? ulen - hi : 0u; // bytes that follow this lane's slice (<= 32768)
                uint32_t moved = gf_mul(c, a.xpow8[after]);         // xpow8[n] = x^(8 n) mod P, built once per context
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) moved ^= (uint32_t)__shfl_xor((int)moved, d);
                const uint32_t crc = ~moved;
                const uint32_t masked = ((crc >> 15) | (crc << 17)) + 0xa282ead8u;
                if (uni(masked) != crc_want) { bad = true; break; }
            }
            if (done + ulen > expect) { bad = true; break; }
            // chunk -> column
            uint8_t *d = dst + done;
            if ((((uintptr_t)d) & 3u) == 0) {
                const uint32_t nd = ulen >> 2;
                for (uint32_t i = lane; i < nd; i += 64) ((uint32_t *)d)[i] = ((const uint32_t *)outb)[i];
                for (uint32_t i = (nd << 2) + lane; i < ulen; i += 64) d[i] = outb[i];
            } else {
                for (uint32_t i = lane; i < ulen; i += 64) d[i] = outb[i];
            }
            done += ulen;
            ip += plen;
            lds_wave_sync();
        }
        if (!bad && done != expect) bad = true;
        any_bad |= bad;
        lds_wave_sync();
    }
    if (__ballot(any_bad) && lane == 0) atomicOr(a.status, 1u);
}

void launch_snappy_sizes(const uint8_t *data, const uint32_t *block_off, int64_t n_blocks, uint32_t *sizes, uint32_t *max_chunk, hipStream_t s) {
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(k_snappy_sizes, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, s, data, block_off, n_blocks, sizes, max_chunk);
}

void launch_snappy_decode(const SnappyArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = 1024 + (size_t)a.in_cap + (size_t)a.out_cap;
    IMM3_LAUNCH_LDS(k_snappy_decode, grid, 64, lds, s, ev0, ev1, a);
}

} // namespace imm3
