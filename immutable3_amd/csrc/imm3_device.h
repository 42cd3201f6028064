// imm3_device.h -- device-side helpers shared by the kernel files (gfx950, wave64).
#pragma once

#include "imm3_internal.h"

namespace imm3 {

typedef int v4i __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------

// x in [lo, hi] (lo <= hi guaranteed by the host) with one subtract and one unsigned compare.
__device__ __forceinline__ bool in_closed(int32_t x, int32_t lo, int32_t hi) {
    return ((uint32_t)x - (uint32_t)lo) <= ((uint32_t)hi - (uint32_t)lo);
}

__device__ __forceinline__ uint64_t ballot64(bool p) { return (uint64_t)__ballot(p); }

// clang has no __builtin_amdgcn_writelane; bind the LLVM intrinsic directly (emits v_writelane_b32).
extern "C" __device__ int imm3_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// A pointer READ FROM MEMORY (a tile table's per-tile column pointer) has no known address space, and every access through it
// becomes a FLAT instruction: slower per access than a global one, and counted by BOTH wait counters -- a wave's LDS waits then
// also wait for its outstanding column loads, which serialises exactly the loads / LDS-transpose overlap the tile kernels are
// built on (round 5: the table instance of k_filter_project ran C3 at 164 us against 122 until its tile pointers went through
// this).  The round trip through address space 1 tells the compiler the pointer is a global one; nothing is emitted.
template <class T>
__device__ __forceinline__ T *as_global(T *p) {
    return (T *)(__attribute__((address_space(1))) T *)p;
}

// mask of the first `rem` bits (rem may be <= 0 or >= 64)
__device__ __forceinline__ uint64_t low_mask(int64_t rem) {
    return rem >= 64 ? ~0ULL : (rem <= 0 ? 0ULL : ((1ULL << rem) - 1ULL));
}

// value of `v` in lane `src_lane` (any lane -> any lane, through the LDS crossbar, no memory)
__device__ __forceinline__ uint32_t lane_read(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

// Move 16 wave-uniform words into lanes 0..15 (lane j receives word j) with v_writelane.
__device__ __forceinline__ uint64_t words_to_lanes(const uint64_t (&acc)[kTileWords]) {
    int lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < kTileWords; ++j) {
        lo = imm3_writelane_i32((int)(uint32_t)acc[j], j, lo);
        hi = imm3_writelane_i32((int)(uint32_t)(acc[j] >> 32), j, hi);
    }
    return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}

// Per-workgroup survivor count -> block_partials[blockIdx.x]; k_total sums them.  (One same-address
// atomicAdd per wave costs ~12 ns serialised: 4096 of them were 40 % of the kernel.)
__device__ __forceinline__ void block_partial_store(uint32_t *block_partials, uint32_t wave_total, int lane, int wave) {
    __shared__ uint32_t s_part[kWavesPerBlock];
    if (lane == 0) s_part[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < kWavesPerBlock; ++i) t += s_part[i];
        block_partials[blockIdx.x] = t;
    }
}

// imm3_query_log_counts: finish[5] = device log (0 = off), finish[6] = next index, finish[7] = capacity.  Called by the
// one thread that produces the count of a run.
__device__ __forceinline__ void count_log_append(unsigned long long *finish, unsigned long long total) {
    unsigned long long *log = (unsigned long long *)finish[5];
    if (!log) return;
    const unsigned long long idx = finish[6];
    if (idx < finish[7]) log[idx] = total;
    finish[6] = idx + 1;
}

// Same, and the count is reduced in the kernel instead of a k_total launch of its own (~4 us of kernel plus a
// dependent-launch gap per scan).  `finish` = {total, n_emit, status, limit, tally, log, log index, log capacity}: every work-group adds
// (its survivors | 1 << 40) to the 64-bit tally with ONE relaxed device-scope atomic -- arrivals in the high bits, the
// running count in the low 40 -- so the group that sees grid - 1 earlier arrivals holds the complete total: no partials
// to re-read, no ordering between two atomics, and no release/acquire fence (a device-scope fence writes back and
// invalidates the XCD's L2: one per work-group made the scan 25 % slower).  One atomic per work-group, spread over the
// kernel's tail, unlike the per-wave atomics of finding 1.
// Work-groups of one launch are equally long and end together, so with a single tally their atomics arrive back to back
// at one address: ~12 ns each, 1536 of them were 7 us of serial tail on a 20 us kernel (which is why round 1 reduced the
// counts of such launches in a k_total launch of its own instead).  The tally is therefore two-level: work-group b adds
// to sub-tally b % kSubTallies (each on a 128-byte line of its own), the last arrival there carries the sub-total to the
// top tally, the last arrival there publishes.  Serial depth grid / 32 + 32 instead of grid.
// one thread per work-group: add this work-group's survivors; the last arrival publishes the total
// (true for the launch's last arrival -- the one that published)
__device__ __forceinline__ bool finish_add(unsigned long long *finish, unsigned long long t) {
    const unsigned int grid = gridDim.x, sub = blockIdx.x % kSubTallies;
    const unsigned int peers = (grid - sub + kSubTallies - 1) / kSubTallies; // work-groups b < grid with b % kSubTallies == sub
    unsigned long long *st = finish + 16 + 16 * sub;
    unsigned long long prev = __hip_atomic_fetch_add(st, t | (1ULL << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((prev >> 40) != (unsigned long long)peers - 1) return false;
    t += prev & ((1ULL << 40) - 1);
    __hip_atomic_store(st, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
    const unsigned int tops = grid < (unsigned int)kSubTallies ? grid : (unsigned int)kSubTallies;
    prev = __hip_atomic_fetch_add(finish + 4, t | (1ULL << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((prev >> 40) == (unsigned long long)tops - 1) {
        const unsigned long long total = (prev & ((1ULL << 40) - 1)) + t;
        const long long limit = (long long)finish[3];
        finish[0] = total;
        finish[1] = (limit > 0 && total > (unsigned long long)limit) ? (unsigned long long)limit : total;
        count_log_append(finish, total);
        finish[kFinishEpoch] += 1; // a new run of the query starts behind this launch (k_filter_project's descriptors)
        __hip_atomic_store(finish + 4, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
        return true;
    }
    return false;
}

// The same for one CHUNK of a limit scan: the launch's last arrival adds the chunk's count to the run's running count
// (finish[kFinishLimitRows]) and publishes THAT as the count so far; a chunk that ran adds its tiles to finish[kFinishLimitTiles].
// The run's first chunk starts both words over (no memset node in front of the run).
__device__ __forceinline__ void finish_add_chunk(unsigned long long *finish, unsigned long long t, unsigned long long tiles_scanned, bool first) {
    const unsigned int grid = gridDim.x, sub = blockIdx.x % kSubTallies;
    const unsigned int peers = (grid - sub + kSubTallies - 1) / kSubTallies;
    unsigned long long *st = finish + 16 + 16 * sub;
    unsigned long long prev = __hip_atomic_fetch_add(st, t | (1ULL << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((prev >> 40) != (unsigned long long)peers - 1) return;
    t += prev & ((1ULL << 40) - 1);
    __hip_atomic_store(st, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int tops = grid < (unsigned int)kSubTallies ? grid : (unsigned int)kSubTallies;
    prev = __hip_atomic_fetch_add(finish + 4, t | (1ULL << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((prev >> 40) == (unsigned long long)tops - 1) {
        const unsigned long long running = (first ? 0ULL : finish[kFinishLimitRows]) + (prev & ((1ULL << 40) - 1)) + t;
        const long long limit = (long long)finish[3];
        finish[kFinishLimitRows] = running;
        finish[kFinishLimitTiles] = (first ? 0ULL : finish[kFinishLimitTiles]) + tiles_scanned;
        finish[0] = running; // (the rows selected in the tiles scanned so far: imm3_query_count runs the whole select when asked)
        finish[1] = (limit > 0 && running > (unsigned long long)limit) ? (unsigned long long)limit : running;
        finish[kFinishEpoch] += 1;
        __hip_atomic_store(finish + 4, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void block_partial_finish_chunk(unsigned long long *finish, uint32_t wave_total, int lane, int wave, unsigned long long tiles_scanned, bool first) {
    __shared__ uint32_t s_part[kWavesPerBlock];
    if (lane == 0) s_part[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int i = 0; i < kWavesPerBlock; ++i) t += s_part[i];
        finish_add_chunk(finish, t, tiles_scanned, first);
    }
}

__device__ __forceinline__ void block_partial_finish(unsigned long long *finish, uint32_t wave_total, int lane, int wave) {
    __shared__ uint32_t s_part[kWavesPerBlock];
    if (lane == 0) s_part[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int i = 0; i < kWavesPerBlock; ++i) t += s_part[i];
        finish_add(finish, t);
    }
}

// LDS hand-off between the lanes of ONE wave: LDS operations of a wave complete in order, so draining lgkmcnt is
// enough (a workgroup-scope fence would also wait for every outstanding global load/store: vmcnt(0)); the asm
// memory clobber keeps the compiler from moving LDS accesses across it.
__device__ __forceinline__ void lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

} // namespace imm3
