// imm3_codec.hip -- PFOR_INT blocks (gfx950, wave64): decode in LDS, fused with the range predicate.
//
// Reference: a PFOR_INT column (core/Column.scala:48,61; dispatched by ScanOp, engine/.../operator/Scan.scala:37-39)
// stores, per storage block, what PFORCodecInt.encode wrote (core/codec/PFORCodec.scala:19-31): the int[] of
// JavaFastPFOR 0.1.10's IntegratedIntCompressor.compress as BIG-endian words, followed by 8 zero bytes:
//     word 0              n = number of values in the block
//     per 4 mini-blocks   header (b1<<24)|(b2<<16)|(b3<<8)|b4, then b1 + b2 + b3 + b4 packed words
//     per leftover one    header b, then b packed words                     (mini-block = 32 values)
//     n % 32 tail values  variable-byte deltas (7 bits per byte, low group first, 0x80 on a value's last byte),
//                         packed little-endian into words
// A mini-block of width b holds 32 wrapping deltas, value i in bits [i*b, (i+1)*b) of its little-endian bit stream;
// b == 32 holds the values themselves, b == 0 nothing.  The delta chain starts at 0 in every storage block.
// The reference's own decode is broken (PFORCodec.scala:43-50 throws on every block); these kernels implement the
// decode its encoder implies (tests/test_gpu_pfor.py checks them against the CPU restatement of that format).
//
// One wave decodes one chunk of up to 1024 values = 32 mini-blocks = 16 values per lane (lane l: mini-block l >> 1,
// values 16 (l & 1) .. +15, i.e. rows 16 l .. 16 l + 15 of the chunk -- the row order the bitmap assembly of the
// int8 tile kernel already uses):
//   1. the chunk's words go HBM -> registers -> LDS with coalesced dword loads, byte-swapped on the way; the loads of
//      the wave's NEXT block are issued before the current one is decoded.  The LDS window is padded by one word per
//      64 (a lane's 16 values start width/2 words after its neighbour's: unpadded, raw mini-blocks would hit 4 banks
//      16 ways), and the pad slot repeats the following word so that every (w, w+1) pair is one ds_read2;
//   2. the <= 8 group headers are walked on the scalar unit (a chain of wave-uniform LDS reads); every mini-block's
//      (first word, width) lands in its lane pair through v_writelane + one ds_bpermute;
//   3. each lane extracts its 16 deltas (ds_read2 + v_alignbit + mask) and sums them locally;
//   4. a segmented wave scan (width-32 mini-blocks restart the chain with absolute values) gives every lane its
//      starting value; the variable-byte tail (last block of a segment only) is decoded by lane 0.
// The kernels are VALU-issue bound, not HBM bound: ~25 vector instructions per 64 values against 3 for a dense int32
// column (DESIGN.md section 12 has the measured rates).
// k_filter_pfor evaluates lo <= v <= hi on the registers and writes the tile's bitmap line: the column is never
// materialised, HBM traffic is the COMPRESSED bytes.  k_pfor_decode writes the values (dense int32 column) for
// everything else (Project, aggregation, ragged layouts, table queries).
#include "imm3_internal.h"
#include "imm3_device.h"
#include <hip/hip_ext.h>

namespace imm3 {

constexpr int kPforWin = 1024 + 8 + 8;                 // logical words per wave window: 32 raw mini-blocks + 8 headers + count + slack
constexpr int kPforRounds = (kPforWin + 63) / 64;      // staging rounds of 64 words
constexpr int kPforLds = 65 * kPforRounds + 2;         // padded LDS words (whole rounds are written)

// LDS slot of logical window word w; slot pw(w) + 1 always holds word w + 1 (see PforRegs::store).
__device__ __forceinline__ int pw(int w) { return w + (w >> 6); }

// A block's words held in registers between the global loads and the LDS writes, so that the loads of the NEXT block
// are in flight while the current one is decoded.  No lane is ever masked off: indices are clamped to the block and
// whole rounds are written (the window has room), which keeps exec-mask bookkeeping off the shared scalar unit.
struct PforRegs {
    uint32_t r[kPforRounds];
    // rounds go in batches of four behind one wave-uniform test (a compressed block is usually 1-2 batches)
    __device__ __forceinline__ void load(const uint32_t *src, int n, int lane) { // n >= 1, wave-uniform
#pragma unroll
        for (int q = 0; q < kPforRounds; q += 4) {
            if (64 * q < n) {
#pragma unroll
                for (int k = q; k < q + 4 && k < kPforRounds; ++k) {
                    const int i = 64 * k + lane;
                    r[k] = __builtin_nontemporal_load(src + (i < n ? i : n - 1));
                }
            }
        }
    }
    __device__ __forceinline__ void store(uint32_t *win, int n, int lane) const {
#pragma unroll
        for (int q = 0; q < kPforRounds; q += 4) {
            if (64 * q < n) {
#pragma unroll
                for (int k = q; k < q + 4 && k < kPforRounds; ++k) win[65 * k + lane] = __builtin_bswap32(r[k]); // pw(64 k + lane)
            }
        }
        lds_wave_sync();
        // the pad slot after word 64 k - 1 repeats word 64 k, so every (w, w + 1) pair is adjacent
        if (lane >= 1 && lane < kPforRounds && 64 * lane < n) win[65 * lane - 1] = win[65 * lane];
        lds_wave_sync();
    }
};

// One step of the segmented inclusive wave scan: (tot, flag) of the DPP source lane is folded in; lanes without a
// source (or in a masked row) fold in (0, 0).  flag is 0 or ~0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step(uint32_t &tot, uint32_t &flag) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tot, CTRL, ROW_MASK, 0xF, true);
    const uint32_t fup = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)flag, CTRL, ROW_MASK, 0xF, true);
    tot += up & ~flag;
    flag |= fup;
}

// Decode `count` (<= 1024, wave-uniform) values whose encoding starts at window word `pos`; `init` is the running
// delta base.  Value 16 * lane + i of the chunk = base + d[i] (garbage where 16 * lane + i >= count).  Returns the
// window position after the chunk, -1 if the block is malformed; `last` = the chunk's last value.
__device__ __forceinline__ int pfor_chunk(const uint32_t *win, int avail, int pos, int count, int32_t init, int lane,
                                          uint32_t *vb, uint32_t (&d)[16], uint32_t &base, int32_t &last) {
    const int n_mini = count >> 5;
    const int n_groups = n_mini >> 2;
    // (1) header walk: a chain of wave-uniform LDS reads.  Per group only the position advances on the scalar unit;
    //     the header word and its position are parked in lane g of two VGPRs and every lane derives its own
    //     mini-block's width and first word from them afterwards.
    int vh = 0, vp = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g < n_groups) {
            const uint32_t hv = win[pw(pos < kPforWin ? pos : kPforWin - 1)];
            const uint32_t hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv);
            vh = imm3_writelane_i32((int)hdr, g, vh);
            vp = imm3_writelane_i32(pos, g, vp);
            const uint32_t sum = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_sad_u8(hv, 0u, 0u)); // b1 + b2 + b3 + b4
            pos += 1 + (int)sum;
        }
    }
    const int m = lane >> 1;
    const int kq = m & 3;
    const uint32_t hdr_m = lane_read((uint32_t)vh, m >> 2);
    const uint32_t pos_m = lane_read((uint32_t)vp, m >> 2);
    uint32_t b = (hdr_m >> (24 - 8 * kq)) & 255u;
    uint32_t o = pos_m + 1u + __builtin_amdgcn_sad_u8((hdr_m >> 8) >> (24 - 8 * kq), 0u, 0u); // + widths of the earlier mini-blocks of the group
    for (int j = n_groups * 4; j < n_mini; ++j) { // one to three leftover mini-blocks, a header word each
        const uint32_t bj = (uint32_t)__builtin_amdgcn_readfirstlane((int)win[pw(pos < kPforWin ? pos : kPforWin - 1)]);
        if (m == j) { b = bj; o = (uint32_t)pos + 1u; }
        pos += 1 + (int)(bj < 256u ? bj : 256u);
    }
    const bool active = m < n_mini;
    if (!active) b = 0;
    const bool wide = b > 32u; // malformed
    b = b > 32u ? 32u : b;
    o = o < (uint32_t)(kPforWin - 34) ? o : (uint32_t)(kPforWin - 34); // o + 32 words + 1 stay inside the window
    const bool raw = b == 32u;
    const uint32_t mask = raw ? ~0u : ((1u << b) - 1u);
    // (2) extraction: value i of this lane sits at bit (16 (lane & 1) + i) * b of the mini-block's stream
    uint32_t bit = (uint32_t)(16 * (lane & 1)) * b;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int w = (int)(o + (bit >> 5));
        const uint32_t *p = win + pw(w);
        d[i] = __builtin_amdgcn_alignbit(p[1], p[0], bit) & mask;
        bit += b;
    }
    // (3) local inclusive sums (raw mini-blocks hold values, not deltas)
    if (!raw) {
#pragma unroll
        for (int i = 1; i < 16; ++i) d[i] += d[i - 1];
    }
    // (4) segmented inclusive scan of the lane totals: a raw lane restarts the chain with its last value
    uint32_t tot = d[15];
    uint32_t flag = raw ? ~0u : 0u; // all-ones once a raw lane has been seen at or below this lane
    if (lane == 0 && !raw) tot += (uint32_t)init;
    seg_step<0x111, 0xF>(tot, flag); // row_shr:1
    seg_step<0x112, 0xF>(tot, flag); // row_shr:2
    seg_step<0x114, 0xF>(tot, flag); // row_shr:4
    seg_step<0x118, 0xF>(tot, flag); // row_shr:8
    seg_step<0x142, 0xA>(tot, flag); // row_bcast:15 -> rows 1, 3
    seg_step<0x143, 0xC>(tot, flag); // row_bcast:31 -> rows 2, 3
    base = (uint32_t)__builtin_amdgcn_update_dpp(init, (int)tot, 0x138, 0xF, 0xF, false); // wave_shr:1; lane 0 keeps init
    if (raw) base = 0;
    uint32_t chain = (uint32_t)init; // the last value decoded so far
    if (n_mini > 0) chain = (uint32_t)__shfl((int)tot, 2 * n_mini - 1);
    int end = pos;
    bool bad = __ballot(wide) != 0 || pos > avail;
    const int n_vb = count & 31;
    if (n_vb) { // wave-uniform: the trailing values of a segment's last block, variable-byte
        bool vbad = false;
        if (lane == 0) {
            int byte = end * 4;
            uint32_t prev = chain;
            for (int k = 0; k < n_vb; ++k) {
                uint32_t val = 0;
                int shift = 0;
                for (;;) {
                    if (byte >= avail * 4 || shift > 28) { vbad = true; break; }
                    const uint32_t c = (win[pw(byte >> 2)] >> (8 * (byte & 3))) & 255u;
                    ++byte;
                    val += (c & 127u) << shift;
                    if (c & 128u) break;
                    shift += 7;
                }
                if (vbad) break;
                prev += val;
                vb[k] = prev;
            }
            end = (byte + 3) >> 2;
            chain = prev;
        }
        lds_wave_sync();
        end = __builtin_amdgcn_readfirstlane(end);
        chain = (uint32_t)__builtin_amdgcn_readfirstlane((int)chain);
        bad |= __builtin_amdgcn_readfirstlane((int)vbad) != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = 16 * lane + i - 32 * n_mini;
            if (k >= 0 && k < n_vb) { d[i] = vb[k]; base = 0; }
        }
        lds_wave_sync();
    }
    last = (int32_t)chain;
    return bad ? -1 : end;
}

// ---------------------------------------------------------------------------------------------
// k_pfor_counts: the value count each block declares (its first word) -- the segment's layout.
// ---------------------------------------------------------------------------------------------
__global__ void k_pfor_counts(const uint8_t *data, const uint32_t *block_off, int64_t n_blocks, int32_t *counts) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_blocks) return;
    const uint32_t o = block_off[k], e = block_off[k + 1];
    counts[k] = e >= o + 4 ? (int32_t)__builtin_bswap32(*(const uint32_t *)(data + o)) : -1;
}

// ---------------------------------------------------------------------------------------------
// k_filter_pfor: tile-aligned layout (block k == bitmap tile k: every block but the last holds 1024 rows).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void k_filter_pfor(const PforArgs a) {
    __shared__ uint32_t s_win[kWavesPerBlock][kPforLds];
    __shared__ uint32_t s_vb[kWavesPerBlock][32];
    // bitmap lines are parked in LDS and stored in bursts of kPark tiles, as in k_filter_tile (stores in between streaming
    // loads cost HBM read/write turnarounds)
    constexpr int kPark = 16;
    __shared__ uint64_t s_park[kWavesPerBlock][kPark][kTileWords]; // 8 KiB
    __shared__ long long s_ptile[kWavesPerBlock][kPark];
    int parked = 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform by construction: tile indices stay in SGPRs
    uint32_t *win = s_win[wave];
    uint32_t lane_total = 0;
    bool any_bad = false;
    const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const uint32_t range = (uint32_t)a.hi - (uint32_t)a.lo;
    PforRegs regs;
    int nw = 0;
    if (wave_id < a.n_tiles) {
        const uint32_t o = a.block_off[wave_id], e = a.block_off[wave_id + 1];
        nw = (int)((e - o) >> 2);
        if (nw >= 1) regs.load((const uint32_t *)(a.data + o), nw < kPforWin ? nw : kPforWin, lane);
    }
    for (int64_t tile = wave_id; tile < a.n_tiles; tile += n_waves) {
        const int staged = nw < kPforWin ? nw : kPforWin;
        regs.store(win, staged, lane);
        bool bad = nw < 1 || nw > kPforWin;
        if (tile + n_waves < a.n_tiles) { // the next block's loads fly while this one is decoded
            const uint32_t o = a.block_off[tile + n_waves], e = a.block_off[tile + n_waves + 1];
            nw = (int)((e - o) >> 2);
            if (nw >= 1) regs.load((const uint32_t *)(a.data + o), nw < kPforWin ? nw : kPforWin, lane);
        }
        const int64_t expect = a.n_rows - tile * kTileRows < kTileRows ? a.n_rows - tile * kTileRows : kTileRows;
        const int count = __builtin_amdgcn_readfirstlane((int)win[0]);
        bad |= count != (int)expect;
        uint32_t d[16], base = 0;
        int32_t last;
        if (!bad) bad = pfor_chunk(win, staged, 1, count, 0, lane, s_vb[wave], d, base, last) < 0;
        any_bad |= bad;
        // lo <= base + d <= hi  <=>  (d + (base - lo)) <=u (hi - lo)
        const uint32_t k = base - (uint32_t)a.lo;
        // the borrow of range - (d + k) is the verdict "outside": shifted in through the carry chain (v_sub_co + v_addc_co per
        // row; a compare would go through a scalar register pair and a v_cndmask), inverted once at the end
        uint32_t bits = 0;
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            uint32_t tmp;
            // (written out: the compiler turns every C form of this into v_cmp + v_cndmask + shift/or with wait states)
            asm("v_sub_co_u32_e32 %1, vcc, %2, %3\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(bits), "=&v"(tmp) : "s"(range), "v"(d[i] + k) : "vcc");
        }
        bits = ~bits & 0xFFFFu;
        if (bad) bits = 0;
        const int src = (lane & 15) << 2; // word j <- lanes 4j .. 4j+3, 16 bits each
        const uint32_t lo = lane_read(bits, src) | (lane_read(bits, src + 1) << 16);
        const uint32_t hi = lane_read(bits, src + 2) | (lane_read(bits, src + 3) << 16);
        uint64_t mine = ((uint64_t)hi << 32) | lo;
        const int64_t w = tile * kTileWords + lane;
        mine &= low_mask(expect - 64 * (int64_t)lane);
        if (lane >= kTileWords) mine = 0;
        if (a.and_existing && lane < kTileWords) mine &= a.bitmap[w];
        if (lane < kTileWords) s_park[wave][parked][lane] = mine; // the bitmap is allocated in whole tiles
        if (lane == 0) s_ptile[wave][parked] = tile;
        lane_total += (uint32_t)__popcll(mine);
        lds_wave_sync(); // the window is reused by the next tile
        if (++parked == kPark || tile + n_waves >= a.n_tiles) { // wave-uniform: burst
            for (int q = lane >> 4; q < parked; q += 4)
                __builtin_nontemporal_store(s_park[wave][q][lane & 15], a.bitmap + s_ptile[wave][q] * kTileWords + (lane & 15));
            lds_wave_sync();
            parked = 0;
        }
    }
#pragma unroll
    for (int dd = 8; dd >= 1; dd >>= 1) lane_total += __shfl_xor(lane_total, dd);
    block_partial_store(a.block_partials, lane_total, lane, wave);
    if (__ballot(any_bad) && lane == 0) atomicOr(a.status, 1u);
}

// ---------------------------------------------------------------------------------------------
// k_pfor_decode: block k -> out[row_base[k] .. row_base[k] + count): any block size, 1024 values per round.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void k_pfor_decode(const PforArgs a) {
    __shared__ uint32_t s_win[kWavesPerBlock][kPforLds];
    __shared__ uint32_t s_vb[kWavesPerBlock][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *win = s_win[wave];
    bool any_bad = false;
    const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t k = wave_id; k < a.n_blocks; k += n_waves) {
        const uint32_t o = a.block_off[k], e = a.block_off[k + 1];
        const int64_t nw = (int64_t)((e - o) >> 2);
        const int64_t expect = (int64_t)a.row_base[k + 1] - (int64_t)a.row_base[k];
        int32_t *out = a.out + a.row_base[k];
        const bool vec = (((uintptr_t)out) & 15) == 0;
        int64_t wpos = 0; // block word the window starts at
        int64_t done = 0;
        int32_t init = 0;
        bool bad = nw < 1;
        while (done < expect && !bad) {
            const int staged = (int)(nw - wpos < kPforWin ? nw - wpos : kPforWin);
            if (staged < 1) { bad = true; break; }
            PforRegs regs;
            regs.load((const uint32_t *)(a.data + o) + wpos, staged, lane);
            regs.store(win, staged, lane);
            int pos = 0;
            if (wpos == 0) {
                bad = (int64_t)(int32_t)win[0] != expect;
                pos = 1;
            }
            const int count = (int)(expect - done < kTileRows ? expect - done : kTileRows);
            uint32_t d[16], base = 0;
            int32_t last = init;
            const int end = bad ? -1 : pfor_chunk(win, staged, pos, count, init, lane, s_vb[wave], d, base, last);
            if (end < 0) { bad = true; break; }
            init = last;
            int32_t *dst = out + done + 16 * lane;
            if (vec && 16 * lane + 16 <= count) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v4i x = {(int)(base + d[4 * q]), (int)(base + d[4 * q + 1]), (int)(base + d[4 * q + 2]), (int)(base + d[4 * q + 3])};
                    *(v4i *)(dst + 4 * q) = x;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (16 * lane + i < count) dst[i] = (int32_t)(base + d[i]);
            }
            done += count;
            wpos += end;
            lds_wave_sync();
        }
        any_bad |= bad;
    }
    if (__ballot(any_bad) && lane == 0) atomicOr(a.status, 1u);
}

void launch_pfor_counts(const uint8_t *data, const uint32_t *block_off, int64_t n_blocks, int32_t *counts, hipStream_t s) {
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(k_pfor_counts, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, s, data, block_off, n_blocks, counts);
}

void launch_filter_pfor(const PforArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    IMM3_LAUNCH(k_filter_pfor, grid, kBlockThreads, s, ev0, ev1, a);
}

void launch_pfor_decode(const PforArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    IMM3_LAUNCH(k_pfor_decode, grid, kBlockThreads, s, ev0, ev1, a);
}

} // namespace imm3
