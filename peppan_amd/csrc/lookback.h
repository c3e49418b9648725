// Decoupled look-back for kernels that SCAN WHILE THEY COMPUTE: a block takes its tile number from a ticket counter (so that a tile only
// ever waits for tiles that are already running), computes its elements, publishes the tile's total and finds the total of everything in
// front of it by looking back over the status words of the tiles before it - the scheme of scan.hip's single-launch scans, as a device
// function, for kernels whose scan input exists only in their registers (selection + compaction in one launch instead of
// flag -> scan -> gather: every launch of a dependent chain costs the GPU 4 - 5 us even when it has next to nothing to do).
//
// status word: epoch << (VB + 2) | flag << VB | value; flag 1 = the tile's own total, 2 = total up to and including the tile.  Words left
// by earlier launches carry another epoch and never match, so nothing is cleared between launches (pep_lookback_begin hands out epochs).
#pragma once
#include "common.h"

constexpr uint64_t LB_SUM = 1, LB_PREFIX = 2;

// ticket of this block (call with all threads; returns the tile number, block-uniform)
__device__ __forceinline__ uint32_t lb_take_tile(uint64_t *state, uint32_t ticket_base, uint32_t *s_tile)
{
    if (threadIdx.x == 0) *s_tile = atomicAdd(reinterpret_cast<uint32_t *>(state), 1u) - ticket_base;
    __syncthreads();
    return *s_tile;
}

// exclusive prefix of tile `tile` whose own total is `tot`.  To be called by the 64 threads of the block's FIRST wavefront (lane = threadIdx.x);
// the result is valid in every one of them.  VB: value bits (32 with a 30-bit epoch, 48 with a 14-bit epoch).
template <int VB>
__device__ __forceinline__ uint64_t lb_tile_prefix(uint64_t *status, uint32_t tile, uint64_t tot, uint64_t epoch, int lane)
{
    constexpr uint64_t VMASK = (1ull << VB) - 1;
    uint64_t prefix = 0;
    if (tile > 0) {
        if (lane == 0) __hip_atomic_store(&status[tile], (epoch << (VB + 2)) | (LB_SUM << VB) | (tot & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int64_t hi = (int64_t)tile - 1; hi >= 0; hi -= 64) {          // a window of 64 predecessors, lane 0 = the nearest
            const int64_t idx = hi - lane;
            uint64_t w = (epoch << (VB + 2)) | (LB_PREFIX << VB);           // in front of tile 0: prefix 0
            if (idx >= 0)
                do { w = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> (VB + 2)) != epoch);
            const uint64_t is_prefix = __ballot(((w >> VB) & 3u) == LB_PREFIX);
            const int stop = is_prefix ? __ffsll((unsigned long long)is_prefix) - 1 : 63;      // lanes 0 .. stop contribute
            uint64_t part = lane <= stop ? (w & VMASK) : 0ull;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
            prefix += part;
            if (is_prefix) break;
        }
    }
    if (lane == 0) __hip_atomic_store(&status[tile], (epoch << (VB + 2)) | (LB_PREFIX << VB) | ((prefix + tot) & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return prefix;
}

// exclusive scan of one value per thread across a 256-thread block; *total = the block's sum (lds: 4 entries)
template <class T>
__device__ __forceinline__ T block_excl_scan_256(T v, T *total, T *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const T o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const T s = lds[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// host side: a launch of n_tiles tiles over the state area S (grown and cleared as needed) -> state pointer (word 0 = ticket counter, status
// words from word 1), the ticket base and the epoch of this launch.  epoch_mask = (1 << epoch bits) - 1 of the word layout used.
int pep_lookback_begin(pep_ctx *ctx, pep_ctx::ScanState &S, uint64_t n_tiles, uint32_t epoch_mask, uint64_t **state, uint32_t *ticket_base, uint64_t *epoch);
