// K2-K4: spaced seeds over a reduced alphabet, bucketed query index, streaming join of the target
// seeds against it, de-duplication of (query, target, diagonal bin) candidates.
//
//  query index            : key of the seed at packed query position p = sum red[r[p+off_k]] * base^k.  Padding bytes (code 31)
//                           never seed, so a seed cannot straddle two sequences.  Index = entries (key << 29 | pos, 8 B) in CSR
//                           buckets by hash(key) + a two-bit Bloom filter (filter_mask).  Built by partition: idx_slab puts the
//                           entries into per-coarse-bucket slabs in one pass, idx_finish sorts every coarse bucket by fine
//                           bucket inside LDS.  seed_count -> scan -> seed_fill (+ filter_fill) is the plain counting sort,
//                           kept for query sets beyond 40 M positions and for coarse buckets that overflow their slab.
//  seed_match             : persistent blocks stream the packed target bytes tile by tile; every position that passes the
//                           filter compares its key with the bucket's entries; equal keys are raw seed hits
//                           (qpos << 32 | tpos), staged in LDS and flushed with one global atomic per ~1.5 k hits.
//  seed_runs_extend       : neighbouring hits of one diagonal share their candidate key q:21 | t:25 | bin:18; every wavefront turns its
//                           slice of the raw hits into runs of equal keys and works them off: already in the device hash set, or
//                           ungapped x-drop extension of the run's hits until one passes -> insert.  All lanes advance their own
//                           extension by sixteen residues per round (no per-residue branching, no runs in global memory).
//  The candidate set is compacted and radix-sorted (sort.hip) so every later stage is order-deterministic.
#include "common.h"
#include <cstring>

namespace {

struct SeedShape {
    int32_t weight;
    int32_t base;
    int32_t offs[32];
    uint32_t red4[4];       // reduced letter of residue code c: nibble (c & 7) of red4[c >> 3]; 15 = never seeds
    int32_t h1;             // the key is assembled as hi * base^h1 + lo with 32-bit halves
    uint32_t pow_h1;
    uint32_t pw[32];        // weight of letter i inside its half: base^i (i < h1), base^(i - h1) otherwise
};

// every shape of a search: the index kernels build all shapes' indices in ONE launch each (blockIdx.y = shape)
struct SeedShapeSet { SeedShape s[4]; };

constexpr int TILE = 256, TILE_HALO = 32;
constexpr int FILTER_SHIFT = 5;              // one 64-bit filter word per 2^FILTER_SHIFT buckets: 5 = 2 bits per bucket (2 MiB at 2^23 buckets, L2-resident); 4 bits per bucket lets 3 % instead of 12 % of the foreign keys through but no longer fits the 4 MiB L2 of an XCD beside the rest: seed_match 0.53 -> 0.59 ms

constexpr uint64_t EMPTY = ~0ull;
constexpr int POS_BITS = 29;
constexpr uint64_t POS_MASK = (1ull << POS_BITS) - 1;

__device__ __forceinline__ uint32_t hash_u64(uint64_t k, int bits)
{
    return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> (64 - bits));
}

// Filter in front of the index ("is this key possibly there?"): a Bloom filter blocked by 64-bit word, two bits per key, 2 bits of
// filter per bucket (2 MiB for 2^23 buckets - it stays in the L2).  word = the top (bucket_bits - 5) bits of the key's hash, so the
// words of one coarse bucket are contiguous; the two bit numbers are the next 6 + 6 bits.  A plain one-bit-per-bucket map lets 33 % of
// the foreign keys through to start[] / entries[] (one 64-byte line each from the memory side), this one 12 %.
__device__ __forceinline__ uint64_t filter_mask(uint64_t k, int bucket_bits, uint32_t &word)
{
    const uint64_t h = k * 0x9E3779B97F4A7C15ull;
    const int wb = bucket_bits - FILTER_SHIFT;
    word = (uint32_t)(h >> (64 - wb));
    return (1ull << ((h >> (58 - wb)) & 63u)) | (1ull << ((h >> (52 - wb)) & 63u));
}

// reduced letters of one 256-position tile (+32 halo) into LDS: one coalesced pass, the reduction table lives in SGPRs
__device__ __forceinline__ uint8_t reduce_letter(const SeedShape &sh, uint32_t code)
{
    const uint32_t c = code & 31u;
    const uint32_t w = (c & 16u) ? ((c & 8u) ? sh.red4[3] : sh.red4[2]) : ((c & 8u) ? sh.red4[1] : sh.red4[0]);
    return (uint8_t)((w >> ((c & 7u) * 4u)) & 15u);
}

// residues of tile `tile` owned by this thread: position x = threadIdx.x and, for the first TILE_HALO threads, the halo byte x + TILE
__device__ __forceinline__ uint32_t fetch_tile(const uint8_t *__restrict__ res, uint64_t tile, uint64_t total)
{
    const uint64_t p = tile * TILE + threadIdx.x;
    uint32_t v = p < total ? res[p] : (uint32_t)PEP_PAD_CODE;
    if (threadIdx.x < TILE_HALO) v |= (uint32_t)(p + TILE < total ? res[p + TILE] : (uint8_t)PEP_PAD_CODE) << 8;
    return v;
}

__device__ __forceinline__ void stage_reduced(const SeedShape &sh, const uint8_t *__restrict__ res, uint64_t tile_base, uint64_t total, uint8_t *red)
{
    for (int x = threadIdx.x; x < TILE + TILE_HALO; x += TILE) {
        const uint64_t p = tile_base + x;
        const uint32_t c = (p < total ? res[p] : (uint8_t)PEP_PAD_CODE) & 31u;
        const uint32_t w = (c & 16u) ? ((c & 8u) ? sh.red4[3] : sh.red4[2]) : ((c & 8u) ? sh.red4[1] : sh.red4[0]);
        red[x] = (uint8_t)((w >> ((c & 7u) * 4u)) & 15u);
    }
}

// key of the seed starting at tile offset x: sum g_i * base^i, assembled from two 32-bit halves.
// W > 0: the weight is a compile-time constant - all letter reads are issued together and every term is one multiply-add with a
// power held in an SGPR (the rolled loop paid one LDS round trip per letter, ~1000 cycles per position);  W = 0: any weight.
template <int W>
__device__ __forceinline__ bool tile_key(const SeedShape &sh, const uint8_t *red, int x, uint64_t &key)
{
    uint32_t lo = 0, hi = 0, mx = 0;
    if (W > 0) {
        constexpr int H1 = (W + 1) / 2;
        uint32_t g[W > 0 ? W : 1];
#pragma unroll
        for (int i = 0; i < W; ++i) g[i] = red[x + sh.offs[i]];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            mx = max(mx, g[i]);
            if (i < H1) lo += g[i] * sh.pw[i]; else hi += g[i] * sh.pw[i];
        }
    } else {
#pragma unroll 1
        for (int i = sh.h1 - 1; i >= 0; --i) { const uint32_t g = red[x + sh.offs[i]]; mx = max(mx, g); lo = lo * (uint32_t)sh.base + g; }
#pragma unroll 1
        for (int i = sh.weight - 1; i >= sh.h1; --i) { const uint32_t g = red[x + sh.offs[i]]; mx = max(mx, g); hi = hi * (uint32_t)sh.base + g; }
    }
    key = (uint64_t)hi * sh.pow_h1 + lo;
    return mx != 15u;                     // 15 = a letter that never seeds (padding, X, ...)
}

template <int W>
__global__ __launch_bounds__(256) void seed_count(SeedShape sh, const uint8_t *__restrict__ res, uint64_t total, uint32_t *__restrict__ cnt, int bucket_bits)
{
    __shared__ uint8_t red[TILE + TILE_HALO];
    stage_reduced(sh, res, (uint64_t)blockIdx.x * TILE, total, red);
    __syncthreads();
    const uint64_t p = (uint64_t)blockIdx.x * TILE + threadIdx.x;
    if (p + 32 > total) return;             // the trailing PEP_END_PAD bytes hold no residues
    uint64_t key;
    if (tile_key<W>(sh, red, threadIdx.x, key)) atomicAdd(&cnt[hash_u64(key, bucket_bits)], 1u);
}

template <int W>
__global__ __launch_bounds__(256) void seed_fill(SeedShape sh, const uint8_t *__restrict__ res, uint64_t total, const uint32_t *__restrict__ start,
                                                 uint32_t *__restrict__ fill, uint64_t *__restrict__ entries, int bucket_bits)
{
    __shared__ uint8_t red[TILE + TILE_HALO];
    stage_reduced(sh, res, (uint64_t)blockIdx.x * TILE, total, red);
    __syncthreads();
    const uint64_t p = (uint64_t)blockIdx.x * TILE + threadIdx.x;
    if (p + 32 > total) return;
    uint64_t key;
    if (tile_key<W>(sh, red, threadIdx.x, key)) {
        const uint32_t b = hash_u64(key, bucket_bits);
        const uint32_t slot = start[b] + atomicAdd(&fill[b], 1u);
        entries[slot] = (key << POS_BITS) | p;
    }
}

// ---- query index by partition (the default build): instead of one memory-side atomic per seed in a count pass and another in a
// fill pass (bound by the ~27 G/s such atomics sustain), the seeds are first split into 2^C coarse buckets with LDS counters
// (one [coarse][block] histogram, scanned once), then every coarse bucket - a few thousand entries - is bucket-sorted by its 2^F
// fine buckets inside LDS by one block, which also writes its slice of start[] and of the occupancy bitmap.
// bucket = coarse << F | fine, C + F = bucket_bits.  A coarse bucket that does not fit LDS (pathologically repetitive input) raises
// a flag and the host rebuilds with the count -> scan -> fill kernels.
constexpr int PART_TILES = 16;                 // 256-position tiles per block in the two partition passes (more for large query sets: <= 2048 blocks)
constexpr int PART_CAP = 5632;                 // capacity of one coarse bucket's slab (an overflow falls back to the plain build)

// One pass over the query positions: a block takes `tiles` tiles in chunks of PART_TILES, keeps the chunk's entries in registers, counts
// them per coarse bucket in LDS, reserves its share of every bucket's slab (PART_CAP entries per coarse bucket) with one global atomic
// per non-empty bucket, and writes the entries there.  (The first version counted in one launch, scanned the [coarse][block] table in a
// second and scattered in a third: 0.035 + 0.025 + 0.052 ms per shape.)  A slab that overflows raises counters[3]: plain build.
template <int W>
__global__ __launch_bounds__(256) void idx_slab(SeedShapeSet shs, const uint8_t *__restrict__ res, uint64_t total, int bucket_bits, int fine_bits,
                                                uint32_t *__restrict__ coarse_cnt, uint64_t *__restrict__ part, uint64_t part_stride, int tiles, uint32_t *__restrict__ counters)
{
    extern __shared__ uint32_t part_lds[];
    // blockIdx.y = shape: one shape's 800 blocks (10 000 genes) are three per CU and wait for memory most of the time - the shapes' builds side by side
    // take the time of one (2 x 0.081 -> 0.10 ms for index build at 10 000 genes)
    const SeedShape &sh = shs.s[blockIdx.y];
    coarse_cnt += (size_t)blockIdx.y * 8192;
    part += (size_t)blockIdx.y * part_stride;
    uint32_t *h = part_lds;                                  // 2^C counters, then write cursors inside the slabs
    const uint32_t n_coarse = 1u << (bucket_bits - fine_bits);
    uint8_t *red = reinterpret_cast<uint8_t *>(h + n_coarse);
    for (int t0 = 0; t0 < tiles; t0 += PART_TILES) {
        for (uint32_t x = threadIdx.x; x < n_coarse; x += 256) h[x] = 0;
        uint64_t ent[PART_TILES];
        uint32_t cb[PART_TILES], fetched[PART_TILES];
        // the residues of the whole chunk are fetched up front (16 loads in flight): a block walks its tiles one after the other with two
        // barriers each, and only ~3 blocks fit a CU's share of the grid - every tile used to start with an exposed memory latency
#pragma unroll
        for (int t = 0; t < PART_TILES; ++t) {
            const uint64_t tile = (uint64_t)blockIdx.x * tiles + t0 + t;
            fetched[t] = (t0 + t < tiles && tile * TILE < total) ? fetch_tile(res, tile, total) : 0u;
        }
#pragma unroll
        for (int t = 0; t < PART_TILES; ++t) {
            const uint64_t base = ((uint64_t)blockIdx.x * tiles + t0 + t) * TILE;
            ent[t] = ~0ull; cb[t] = 0;
            __syncthreads();
            if (t0 + t < tiles && base < total) {           // block-uniform
                red[threadIdx.x] = reduce_letter(sh, fetched[t]);
                if (threadIdx.x < TILE_HALO) red[TILE + threadIdx.x] = reduce_letter(sh, fetched[t] >> 8);
                __syncthreads();
                const uint64_t p = base + threadIdx.x;
                uint64_t key;
                if (p + 32 <= total && tile_key<W>(sh, red, threadIdx.x, key)) {
                    ent[t] = (key << POS_BITS) | p;
                    cb[t] = hash_u64(key, bucket_bits) >> fine_bits;
                    atomicAdd(&h[cb[t]], 1u);
                }
            }
        }
        __syncthreads();
        // slab space of this chunk: one returning global atomic per coarse bucket - eight of a thread's are in flight together (issued one
        // by one, each waiting for its answer, they were a third of the kernel)
        for (uint32_t x0 = threadIdx.x; x0 < n_coarse; x0 += 8 * 256) {
            uint32_t c8[8], at8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const uint32_t x = x0 + 256u * k; c8[k] = x < n_coarse ? h[x] : 0u; }
#pragma unroll
            for (int k = 0; k < 8; ++k) at8[k] = x0 + 256u * k < n_coarse ? atomicAdd(&coarse_cnt[x0 + 256u * k], c8[k]) : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c8[k]) {
                    if (at8[k] + c8[k] > (uint32_t)PART_CAP) counters[3] = 1u;
                    h[x0 + 256u * k] = at8[k];
                }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < PART_TILES; ++t)
            if (ent[t] != ~0ull) {
                const uint32_t slot = atomicAdd(&h[cb[t]], 1u);
                if (slot < (uint32_t)PART_CAP) part[(uint64_t)cb[t] * PART_CAP + slot] = ent[t];
            }
        __syncthreads();
    }
}

#ifdef PEP_PROBES
// MEASUREMENT ONLY, compiled with -DPEP_PROBES (make PROBES=1; params.reserved[0] = 7, tools/partition_probe.py): the first pass of a PARTITIONED join - the target positions whose key
// passes the filter are scattered into the coarse buckets of the query index, exactly as idx_slab scatters the query positions (same chunking,
// same LDS counting, same slab reservation), 8 bytes (key << 29 | position) each.  Its duration against seed_match's is what decides whether a
// partitioned join can pay; nothing reads what it writes.
template <int W>
__global__ __launch_bounds__(256) void tgt_slab_probe(SeedShape sh, const uint8_t *__restrict__ res, uint64_t total, int bucket_bits, int fine_bits,
                                                      const unsigned long long *__restrict__ filter, uint32_t *__restrict__ coarse_cnt, uint64_t *__restrict__ part,
                                                      uint32_t cap, int tiles, unsigned long long *__restrict__ kept)
{
    extern __shared__ uint32_t part_lds[];
    uint32_t *h = part_lds;
    const uint32_t n_coarse = 1u << (bucket_bits - fine_bits);
    uint8_t *red = reinterpret_cast<uint8_t *>(h + n_coarse);
    unsigned long long mine = 0;
    for (int t0 = 0; t0 < tiles; t0 += PART_TILES) {
        for (uint32_t x = threadIdx.x; x < n_coarse; x += 256) h[x] = 0;
        uint64_t ent[PART_TILES];
        uint32_t cb[PART_TILES], fetched[PART_TILES];
#pragma unroll
        for (int t = 0; t < PART_TILES; ++t) {
            const uint64_t tile = (uint64_t)blockIdx.x * tiles + t0 + t;
            fetched[t] = (t0 + t < tiles && tile * TILE < total) ? fetch_tile(res, tile, total) : 0u;
        }
#pragma unroll
        for (int t = 0; t < PART_TILES; ++t) {
            const uint64_t base = ((uint64_t)blockIdx.x * tiles + t0 + t) * TILE;
            ent[t] = ~0ull; cb[t] = 0;
            __syncthreads();
            if (t0 + t < tiles && base < total) {
                red[threadIdx.x] = reduce_letter(sh, fetched[t]);
                if (threadIdx.x < TILE_HALO) red[TILE + threadIdx.x] = reduce_letter(sh, fetched[t] >> 8);
                __syncthreads();
                const uint64_t p = base + threadIdx.x;
                uint64_t key;
                if (p + 32 <= total && tile_key<W>(sh, red, threadIdx.x, key)) {
                    uint32_t word;
                    const uint64_t m = filter_mask(key, bucket_bits, word);
                    if ((filter[word] & m) == m) {
                        ent[t] = (key << POS_BITS) | p;
                        cb[t] = hash_u64(key, bucket_bits) >> fine_bits;
                        atomicAdd(&h[cb[t]], 1u);
                        ++mine;
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t x0 = threadIdx.x; x0 < n_coarse; x0 += 8 * 256) {
            uint32_t c8[8], at8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const uint32_t x = x0 + 256u * k; c8[k] = x < n_coarse ? h[x] : 0u; }
#pragma unroll
            for (int k = 0; k < 8; ++k) at8[k] = x0 + 256u * k < n_coarse ? atomicAdd(&coarse_cnt[x0 + 256u * k], c8[k]) : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (c8[k]) h[x0 + 256u * k] = at8[k];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < PART_TILES; ++t)
            if (ent[t] != ~0ull) {
                const uint32_t slot = atomicAdd(&h[cb[t]], 1u);
                if (slot < cap) part[(uint64_t)cb[t] * cap + slot] = ent[t];
            }
        __syncthreads();
    }
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(kept, mine);
}
#endif  // PEP_PROBES

// one block per coarse bucket c: its slab -> entries[] ordered by fine bucket (dense: after the entries of the coarse buckets before it),
// start[c << F .. (c + 1) << F), filter slice
__global__ __launch_bounds__(256) void idx_finish(const uint64_t *__restrict__ part, const uint32_t *__restrict__ coarse_cnt, int bucket_bits,
                                                  int fine_bits, uint32_t *__restrict__ start, uint64_t *__restrict__ entries,
                                                  unsigned long long *__restrict__ filter, uint32_t *__restrict__ counters, uint32_t *__restrict__ n_entries_out,
                                                  uint64_t part_stride, uint64_t start_stride, uint64_t entries_stride, uint64_t filter_stride)
{
    __shared__ uint32_t pos[4096];
    part += (size_t)blockIdx.y * part_stride;                  // blockIdx.y = shape (see idx_slab)
    coarse_cnt += (size_t)blockIdx.y * 8192;
    start += (size_t)blockIdx.y * start_stride;
    entries += (size_t)blockIdx.y * entries_stride;
    filter += (size_t)blockIdx.y * filter_stride;
    n_entries_out += blockIdx.y;
    __shared__ unsigned long long fw[4096 >> FILTER_SHIFT];
    __shared__ uint32_t wave_sum[4];
    const uint32_t c = blockIdx.x, n_coarse = 1u << (bucket_bits - fine_bits), n_fine = 1u << fine_bits;
    // entries before this coarse bucket: every block adds up the (few thousand) counts ahead of it
    uint32_t before = 0;
    for (uint32_t x = threadIdx.x; x < c; x += 256) before += coarse_cnt[x];
    for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = before;
    __syncthreads();
    const uint32_t lo = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    const uint32_t n = coarse_cnt[c];
    __syncthreads();                                           // wave_sum is used again below
    if (c == n_coarse - 1 && threadIdx.x == 0) { start[(size_t)n_coarse << fine_bits] = lo + n; *n_entries_out = lo + n; }      // (the second copy: read back with the search's counters)
    if (n > PART_CAP) {
        // the host falls back to count -> scan -> fill, but the matcher of THIS attempt still runs: leave it well-formed (empty) buckets and
        // filter words instead of whatever the buffers held before
        if (threadIdx.x == 0) counters[3] = 1u;
        for (uint32_t x = threadIdx.x; x < n_fine; x += 256) start[((size_t)c << fine_bits) + x] = lo;
        for (uint32_t x = threadIdx.x; x < (n_fine >> FILTER_SHIFT); x += 256) filter[((size_t)c << (fine_bits - FILTER_SHIFT)) + x] = 0ull;
        return;
    }
    for (uint32_t x = threadIdx.x; x < n_fine; x += 256) pos[x] = 0;
    for (uint32_t x = threadIdx.x; x < (n_fine >> FILTER_SHIFT); x += 256) fw[x] = 0;
    __syncthreads();
    const uint32_t fmask = n_fine - 1;
    // the slab is read twice (count, place) from the L2 instead of being staged: 45 KiB of LDS less, four times the resident blocks
    const uint64_t *slab = part + (uint64_t)c * PART_CAP;
    for (uint32_t x = threadIdx.x; x < n; x += 256) {
        const uint64_t e = slab[x];
        atomicAdd(&pos[hash_u64(e >> POS_BITS, bucket_bits) & fmask], 1u);
        uint32_t word;
        const uint64_t m = filter_mask(e >> POS_BITS, bucket_bits, word);
        atomicOr(&fw[word & ((n_fine >> FILTER_SHIFT) - 1)], (unsigned long long)m);
    }
    __syncthreads();
    // exclusive scan of the fine counters: thread t owns n_fine / 256 consecutive counters
    const uint32_t per = n_fine / 256;                       // fine_bits >= 8
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; ++k) sum += pos[threadIdx.x * per + k];
    uint32_t incl = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wave_sum[w];
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t f = threadIdx.x * per + k, cnt = pos[f];
        pos[f] = run;
        run += cnt;
    }
    __syncthreads();
    // start[] of this coarse bucket from the scanned counters, one coalesced pass (a thread writing its own `per` consecutive words
    // touches every 64-byte line of the slice `per` times)
    for (uint32_t x = threadIdx.x; x < n_fine; x += 256) start[((size_t)c << fine_bits) + x] = lo + pos[x];
    // this coarse bucket's slice of the filter: 2^(fine_bits - 5) words
    for (uint32_t x = threadIdx.x; x < (n_fine >> FILTER_SHIFT); x += 256) filter[((size_t)c << (fine_bits - FILTER_SHIFT)) + x] = fw[x];
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < n; x += 256) {
        const uint64_t e = slab[x];
        const uint32_t slot = atomicAdd(&pos[hash_u64(e >> POS_BITS, bucket_bits) & fmask], 1u);
        entries[lo + slot] = e;
    }
}

// the filter for an index built the plain way (count -> scan -> fill): every entry sets its two bits (filter zeroed by the caller)
__global__ __launch_bounds__(256) void filter_fill(const uint64_t *__restrict__ entries, const uint32_t *__restrict__ n_entries, int bucket_bits,
                                                   unsigned long long *__restrict__ filter, uint32_t *__restrict__ n_entries_out)
{
    const uint32_t n = *n_entries;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_entries_out = n;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256) {
        uint32_t word;
        const uint64_t m = filter_mask(entries[e] >> POS_BITS, bucket_bits, word);
        atomicOr(&filter[word], (unsigned long long)m);
    }
}

struct JoinArgs {
    const uint8_t *t_res;
    uint64_t t_total;
    const uint32_t *t_off;
    uint32_t nt;
    const uint32_t *q_off;
    uint32_t nq;
    const uint2 *q_blk2seq, *t_blk2seq;      // (sequence, its start) per 32-byte block of the packed sets (half the table of round 3: 4.2 MiB instead of 8.4 at 50 000 genes -
                                             // seed_runs is bound by the misses of exactly this look-up, profiles/r04_seed_counters_50k.txt)
    const uint32_t *start;
    const uint64_t *entries;
    const unsigned long long *filter;     // filter_mask(): two bits per key in one 64-bit word (2 MiB for 2^23 buckets: stays in L2, unlike start[])
    int bucket_bits;
    uint64_t *table;
    int table_bits;
    uint32_t *counters;      // [0] = list length (filled by set_compact), [1] = set overflow flag, [2] = hit buffer overflow
    unsigned long long *stats;   // [0] = target seeds, [1] = seed hits, [2] = seed hits passing the ungapped filter
    const uint8_t *q_res;
    const int8_t *sub;       // 32x32 substitution scores (global; staged in LDS by the kernel)
    int ungapped_min, xdrop, ext_right, ext_left, stage1_min;
    uint64_t *hits;          // raw seed hits (qpos << 32 | tpos)
    unsigned long long *hit_count;
    uint64_t hit_cap;
    int debug;               // profiling aid (params.reserved[0]): 1 = keys only, 2 = keys + bucket lookup, 3 = + entry compare (no extension), 9 = no wave-level de-duplication
    unsigned int *tile_ctr;      // the matchers' tile claims (TileClaims): tile_groups counters, 128 bytes apart, zero when the launch starts
    uint32_t tile_groups;        // 8 (a launch of 64 blocks and more: block b claims from counter b mod 8 - neighbouring blocks go to different XCCs, a counter a group keeps the
                                 // atomics of one word at a tenth of what it sustains) or 1
    const int32_t *self_delta;   // self-search (pep_self_map): per 32-byte block of the targets, target position - query position of the hits that are a gene against itself on
                                 // diagonal 0 (PEP_SELF_NO_DELTA: none); nullptr = off.  Such hits are counted and dropped: self_prepare has settled their candidate.
};

__device__ __forceinline__ void set_insert(const JoinArgs &a, uint64_t k)
{
    const uint32_t mask = (1u << a.table_bits) - 1;
    uint32_t slot = hash_u64(k, a.table_bits);
    for (uint32_t probe = 0; probe < 256 && probe <= mask; ++probe) {
        const uint64_t cur = a.table[slot];
        if (cur == k) return;
        if (cur == EMPTY) {
            const uint64_t old = atomicCAS((unsigned long long *)&a.table[slot], (unsigned long long)EMPTY, (unsigned long long)k);
            if (old == k) return;
            if (old == EMPTY) return;
        }
        slot = (slot + 1) & mask;
    }
    a.counters[1] = 1u;
}

__device__ __forceinline__ bool set_contains(const JoinArgs &a, uint64_t k)
{
    const uint32_t mask = (1u << a.table_bits) - 1;
    uint32_t slot = hash_u64(k, a.table_bits);
    for (uint32_t probe = 0; probe < 256 && probe <= mask; ++probe) {
        const uint64_t cur = a.table[slot];
        if (cur == k) return true;
        if (cur == EMPTY) return false;          // possibly stale: the caller then extends and inserts with a CAS
        slot = (slot + 1) & mask;
    }
    return false;
}

// (the ungapped x-drop extension itself - BLOSUM62 along the diagonal through a seed hit, on PACKED positions: the >= 16 padding bytes around every
// sequence score -64, which ends an extension exactly where the sequence ends (x-drop < 64), so no bounds are needed - lives in seed_extend below)

// Which tiles a block of a matcher works on: claims of TILE_CLAIM consecutive tiles, handed out by atomic counters - group g of the blocks (block number mod tile_groups)
// owns the claims g, g + tile_groups, ... and counts through them.  A claim is asked for one claim AHEAD (when the block starts the claim in front) and looked at when that
// claim's last tile has been staged, so the atomic's round trip is never waited for alone.  Every member function is called by all threads of the block.
// (TILE_CLAIM: four tiles of 256 positions for seed_match - 1, 2, 4, 8 take the same time -, two of the stride matcher's tiles of 1 024: a block has ten of those in all at
// 10 000 genes - 0.167 ms with 2, 0.178 with 4, 0.193 with 8; profiles/r06_block_times.txt)
template <uint32_t TILE_CLAIM>
struct TileClaims {
    const JoinArgs &a;
    const uint64_t n_tiles;
    const uint32_t grp;
    unsigned int *ctr;
    uint32_t ahead = 0;              // thread 0: the answer to the claim that is in flight
    uint64_t end = 0;                // one behind the last tile of the claim the block is in
    bool asked = false;
    uint32_t turn = 0;               // which of the two slots the claim behind this one is handed over in (two: a slot is written again only after a barrier that every reader of its last value has passed)
    __device__ TileClaims(const JoinArgs &args, uint64_t tiles) : a(args), n_tiles(tiles), grp(args.tile_groups > 1 ? blockIdx.x % args.tile_groups : 0u), ctr(args.tile_ctr + 32u * grp) {}
    __device__ uint64_t first_tile_of(uint32_t j) const { return ((uint64_t)j * a.tile_groups + grp) * TILE_CLAIM; }
    __device__ uint64_t first()
    {
        __shared__ uint32_t s_first;
        if (threadIdx.x == 0) s_first = atomicAdd(ctr, 1u);
        __syncthreads();
        const uint64_t t = first_tile_of(s_first);
        end = min(t + TILE_CLAIM, n_tiles);
        return t;
    }
    // in front of the barrier that follows the staging of `tile`
    __device__ void before_barrier(uint64_t tile)
    {
        if (!asked) { if (threadIdx.x == 0) ahead = atomicAdd(ctr, 1u); asked = true; }
        if (tile + 1 == end && threadIdx.x == 0) slot()[turn] = ahead;
    }
    // behind that barrier: the tile that follows `tile` for this block (>= n_tiles: none)
    __device__ uint64_t next(uint64_t tile)
    {
        if (tile + 1 != end) return tile + 1;
        const uint64_t t = first_tile_of(slot()[turn]);
        end = min(t + TILE_CLAIM, n_tiles);
        asked = false;
        turn ^= 1u;
        return t;
    }
    __device__ static uint32_t *slot() { __shared__ uint32_t s_claim[2]; return s_claim; }
};

// Phase 1 of the join (persistent over 256-position tiles, claimed four at a time): target seed keys are looked up in the
// query index and every equal-key pair is appended as a raw seed hit (qpos << 32 | tpos).  Hits are staged in an
// LDS buffer and flushed with ONE global atomic per flush (a single global counter word only sustains ~90
// atomics/us).  Every memory operation of this phase is independent across lanes: high memory-level parallelism.
#ifndef PEP_HIT_BUF
#define PEP_HIT_BUF 2048
#endif
#ifdef PEP_PROBES
// MEASUREMENT ONLY (make PROBES=1; tools/ab/block_times.py): when and where every block of the last seed_match launch ran - start and end on the device's wall clock,
// the hardware id of the wavefront that wrote them (compute unit, shader engine, XCC)
__device__ unsigned long long g_block_probe[4 * 2048];
#endif
#ifndef PEP_ENTRY_TRIP
#define PEP_ENTRY_TRIP 4
#endif
#ifndef PEP_STRIDE_ENTRY_TRIP
#define PEP_STRIDE_ENTRY_TRIP 2
#endif
constexpr int STRIDE_ENTRY_TRIP = PEP_STRIDE_ENTRY_TRIP;      // ... and of seed_match_stride's
constexpr int ENTRY_TRIP = PEP_ENTRY_TRIP;      // entries of a bucket per trip of seed_match's walk (make EXTRA=-DPEP_ENTRY_TRIP=4: a measurement build)
constexpr int HIT_BUF = PEP_HIT_BUF;          // (make EXTRA=-DPEP_HIT_BUF=1024: a measurement build - how many of the matcher's blocks a CU holds at once is a matter of this buffer)
template <int W>
__global__ __launch_bounds__(256) void seed_match(SeedShape sh, JoinArgs a)
{
    __shared__ uint64_t buf[HIT_BUF];
    __shared__ uint8_t red[TILE + TILE_HALO];
    __shared__ uint32_t nbuf, blk_stats[2];
    __shared__ unsigned long long gbase;
    if (threadIdx.x == 0) { nbuf = 0; blk_stats[0] = blk_stats[1] = 0; }
#ifdef PEP_PROBES
    if (threadIdx.x == 0 && blockIdx.x < 2048) {
        g_block_probe[4 * blockIdx.x] = wall_clock64();
        g_block_probe[4 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20) << 32) | __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // XCC_ID | HW_ID
    }
#endif
    __syncthreads();
    uint32_t n_seed = 0, n_hit = 0;
    const uint64_t n_tiles = (a.t_total + 255) / 256;
    // The tiles are CLAIMED, TILE_CLAIM consecutive ones at a time (TileClaims above), not dealt out by block number: the launch asks for as many blocks as the chip
    // holds at once (eight per compute unit), but which unit a block lands on is the dispatcher's business - on most boxes of the pool it gives some units nine to
    // eleven of these blocks and others seven, and the blocks beyond a unit's eighth start when its first ones END.  With a fixed share of the tiles per block those
    // late blocks (an eighth of all) made the launch half as long again (324 us against 183 us for the blocks that started at once, profiles/r06_block_times.txt);
    // a block that starts late now finds what is left.
    TileClaims<4> claims(a, n_tiles);
    uint64_t tile = claims.first();
    // the residues of the NEXT tile are fetched into a register while this one is processed: a block walks ~40 tiles one after the
    // other, and without this every tile starts with an exposed global-memory latency
    uint32_t fetched = tile < n_tiles ? fetch_tile(a.t_res, tile, a.t_total) : 0u;
    // (self-search: the block's distance word rides along with the residues - eight distinct words per tile, one request per wavefront)
    int32_t sd_next = (a.self_delta && tile < n_tiles) ? a.self_delta[(tile * 256 + threadIdx.x) >> 5] : PEP_SELF_NO_DELTA;
    while (tile < n_tiles) {
        const uint64_t p = tile * 256 + threadIdx.x;
        uint64_t key = 0;
        uint32_t e0 = 0, e1 = 0;
        const int32_t sd = sd_next;
        red[threadIdx.x] = reduce_letter(sh, fetched);
        if (threadIdx.x < TILE_HALO) red[TILE + threadIdx.x] = reduce_letter(sh, fetched >> 8);
        claims.before_barrier(tile);
        __syncthreads();
        const uint64_t next_tile = claims.next(tile);          // (block-uniform; >= n_tiles: this is the block's last tile)
        if (next_tile < n_tiles) {
            fetched = fetch_tile(a.t_res, next_tile, a.t_total);
            if (a.self_delta) sd_next = a.self_delta[(next_tile * 256 + threadIdx.x) >> 5];
        }
        if (p + 32 <= a.t_total && tile_key<W>(sh, red, threadIdx.x, key)) {
            ++n_seed;
            if (a.debug != 1) {
                const uint32_t b = hash_u64(key, a.bucket_bits);
                uint32_t word;
                const uint64_t m = filter_mask(key, a.bucket_bits, word);
                // (a position inside a stretch that repeats a query holds that query's key: it is in the index, the filter's word need not be asked - the filter only ever
                // saves look-ups, so a block that is marked beyond the stretch's end costs a look-up and nothing else; debug 11: ask anyway, for comparison)
                if ((sd != PEP_SELF_NO_DELTA && a.debug != 11) || (a.filter[word] & m) == m) {
                    // (both ends of the bucket with ONE 8-byte request - the two words lie side by side, and what this kernel pays for is requests; debug 12: two loads, for comparison)
                    if (a.debug == 12) { e0 = a.start[b]; e1 = a.start[b + 1]; }
                    else { uint32_t ends[2]; __builtin_memcpy(ends, a.start + b, 8); e0 = ends[0]; e1 = ends[1]; }
                }
                if (a.debug == 2) { n_hit += e1 - e0; e1 = e0; }
            }
        }
        // two entries per trip (one unaligned 16-byte load): a bucket that holds the key usually holds one to four entries - the members of a
        // gene family - and every trip of this loop is a dependent round trip to the L2 (the entry behind the bucket's last one is read and
        // ignored; the array has a spare slot)
        for (uint32_t e = e0; e < e1; e += ENTRY_TRIP) {
            uint64_t pair[ENTRY_TRIP];
            __builtin_memcpy(pair, a.entries + e, 8 * ENTRY_TRIP);
#pragma unroll
            for (int k = 0; k < ENTRY_TRIP; ++k) {
                const uint64_t ent = pair[k];
                if (e + k >= e1 || (ent >> POS_BITS) != key) continue;
                ++n_hit;
                if (a.debug == 3) continue;
                if ((int32_t)((uint32_t)p - (uint32_t)(ent & POS_MASK)) == sd) continue;      // a gene against itself on diagonal 0: settled by self_prepare
                const uint64_t hit = ((ent & POS_MASK) << 32) | p;
                const uint32_t idx = atomicAdd(&nbuf, 1u);
                if (idx < HIT_BUF) buf[idx] = hit;
                else {                                   // staging buffer full: rare direct append
                    const unsigned long long g = atomicAdd(a.hit_count, 1ull);
                    if (g < a.hit_cap) a.hits[g] = hit; else a.counters[2] = 1u;
                }
            }
        }
        __syncthreads();
        const uint32_t cnt = nbuf;                   // block-uniform after the barrier
        if (cnt > HIT_BUF / 2 || next_tile >= n_tiles) {
            const uint32_t n = min(cnt, (uint32_t)HIT_BUF);
            if (threadIdx.x == 0) gbase = n ? atomicAdd(a.hit_count, (unsigned long long)n) : 0ull;
            __syncthreads();
            const unsigned long long g = gbase;
            for (uint32_t x = threadIdx.x; x < n; x += 256) {
                if (g + x < a.hit_cap) a.hits[g + x] = buf[x]; else a.counters[2] = 1u;
            }
            __syncthreads();
            if (threadIdx.x == 0) nbuf = 0;
            __syncthreads();
        }
        tile = next_tile;
    }
    for (int d = 32; d > 0; d >>= 1) {
        n_seed += __shfl_down(n_seed, d, 64);
        n_hit += __shfl_down(n_hit, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&blk_stats[0], n_seed); atomicAdd(&blk_stats[1], n_hit); }
    __syncthreads();
    if (threadIdx.x < 2 && blk_stats[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)blk_stats[threadIdx.x]);
#ifdef PEP_PROBES
    if (threadIdx.x == 0 && blockIdx.x < 2048) g_block_probe[4 * blockIdx.x + 1] = wall_clock64();
#endif
}

#ifdef PEP_PROBES
extern "C" int pep_probe_block_times(unsigned long long *out)          // 4 words per block of the last seed_match launch: start, end, XCC_ID << 32 | HW_ID, unused
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_probe), sizeof(g_block_probe)) == hipSuccess ? 0 : -1;
}
#endif

// ---- the nucleotide tool's matcher (round 6): BLAST's lookup stride (blastn -word_size 17 looks up shorter words at a stride and verifies them,
// uberBlast.py:294).  The seeds are exact NW-mers over the four bases.  Every NW-mer holds exactly one NK-mer that starts at a packed position
// = 0 (mod NS), NS = NW - NK + 1: the query index holds the NK-mers of EVERY query position, the targets are probed at every NS-th position only,
// and a probe that finds its NK-mer at query position q looks at the NS - 1 bases on either side (the target's are in the tile, the query's are one
// 4-byte read per side): with L matching bases to the left and R to the right (each capped at NS - 1) the NW-mers that start k = NS-1-R .. L bases
// in front are exactly the seed hits this probe owns.  The raw hits are the plain matcher's, every one of them once - the candidate set, and with it
// the hit table, is bit-identical - at a quarter of the probes: the plain matcher pays filter word + start[] + entries for EVERY position of an exact
// run (three scattered requests per raw hit), this one pays them once per four.  NK = 14: 4^14 keys against ~10 M query positions keeps the filter
// selective (3.7 % of the keys occupied; NK = 12 / stride 6 has 60 % occupied: every probe would walk a bucket and verify 0.6 chance matches).
constexpr int NK = 14, NS = 4, NW = 17;
constexpr int NTILE = 256 * NS;                       // target positions per tile: one probe per thread
constexpr uint32_t PAD4 = 0x01010101u * PEP_PAD_CODE;

__device__ __forceinline__ uint32_t nt_pack4(uint32_t d) { return (d | (d >> 6) | (d >> 12) | (d >> 18)) & 0xFFu; }          // four codes < 4 -> 8 bits, first base lowest (a byte >= 4 would spill into its neighbours' bits)
__device__ __forceinline__ uint32_t nt_bad4(uint32_t d)                                                                    // bit i: byte i is not one of the four bases
{
    const uint32_t t = ((d >> 2) | (d >> 3) | (d >> 4)) & 0x01010101u;
    return ((t * 0x01020408u) >> 24) & 0xFu;
}

// dword g of the packed targets (4-byte aligned: the buffers are hipMalloc'ed); padding outside [0, total)
__device__ __forceinline__ uint32_t nt_fetch(const uint8_t *__restrict__ res, int64_t g, uint64_t total)
{
    if (g < 0) return PAD4;
    const uint64_t b = (uint64_t)g * 4;
    if (b + 4 <= total) return reinterpret_cast<const uint32_t *>(res)[g];
    uint32_t v = PAD4;
    for (int k = 0; k < 4; ++k)
        if (b + k < total) v = (v & ~(0xFFu << (8 * k))) | ((uint32_t)res[b + k] << (8 * k));
    return v;
}

__global__ __launch_bounds__(256) void seed_match_stride(JoinArgs a)
{
    static_assert(NK + NS - 1 == NW && NS == 4 && NK == 14, "one dword of flank per side, the look-up word in four dwords");
    __shared__ uint64_t buf[HIT_BUF];
    __shared__ uint32_t win[2][NTILE / 4 + 8];        // bytes [base - 4, base + NTILE + 28) of the tile, as dwords; two tiles in turn (one barrier per tile)
    __shared__ uint32_t nbuf, blk_stats[2];
    __shared__ unsigned long long gbase;
    if (threadIdx.x == 0) { nbuf = 0; blk_stats[0] = blk_stats[1] = 0; }
    __syncthreads();
    uint32_t n_seed = 0, n_hit = 0;
    const uint64_t n_tiles = (a.t_total + NTILE - 1) / NTILE;
    const int x = threadIdx.x;
    // this thread's dwords of the NEXT tile are fetched while the current one is worked on
    uint32_t f0 = 0, f1 = 0;
    int32_t sd_next = PEP_SELF_NO_DELTA;             // self-search: the distance word of this thread's probe position rides along (JoinArgs::self_delta)
    TileClaims<2> claims(a, n_tiles);                // (the tiles are claimed, two at a time, as seed_match's are: a block that the dispatcher starts late finds what is left)
    uint64_t tile = claims.first();
    if (tile < n_tiles) {
        const int64_t g0 = (int64_t)tile * (NTILE / 4) - 1;
        f0 = nt_fetch(a.t_res, g0 + x, a.t_total);
        if (x < 8) f1 = nt_fetch(a.t_res, g0 + 256 + x, a.t_total);
        if (a.self_delta) sd_next = a.self_delta[(tile * NTILE + 4u * x) >> 5];
    }
    for (int turn = 0; tile < n_tiles; turn ^= 1) {
        uint32_t *w = win[turn];
        w[x] = f0;
        if (x < 8) w[256 + x] = f1;
        const int32_t sd = sd_next;
        claims.before_barrier(tile);
        __syncthreads();
        const uint64_t next_tile = claims.next(tile);          // (block-uniform; >= n_tiles: this is the block's last tile)
        if (next_tile < n_tiles) {
            const int64_t g0 = (int64_t)next_tile * (NTILE / 4) - 1;
            f0 = nt_fetch(a.t_res, g0 + x, a.t_total);
            if (x < 8) f1 = nt_fetch(a.t_res, g0 + 256 + x, a.t_total);
            if (a.self_delta) sd_next = a.self_delta[(next_tile * NTILE + 4u * x) >> 5];
        }
        const uint32_t p = (uint32_t)(tile * NTILE) + 4u * x;            // the probe: packed target position, = 0 (mod NS)
        const uint32_t d0 = w[x], d1 = w[x + 1], d2 = w[x + 2], d3 = w[x + 3], d4 = w[x + 4], d5 = w[x + 5];      // bytes p - 4 .. p + 19
        const uint32_t inv = nt_bad4(d1) | (nt_bad4(d2) << 4) | (nt_bad4(d3) << 8) | (nt_bad4(d4) << 12) | (nt_bad4(d5) << 16);
        // (statistics: target positions that start an NW-mer of the four bases)
#pragma unroll
        for (int j = 0; j < NS; ++j) n_seed += ((inv >> j) & ((1u << NW) - 1)) == 0 ? 1u : 0u;
        uint32_t e0 = 0, e1 = 0;
        uint64_t key = 0;
        if ((inv & ((1u << NK) - 1)) == 0) {
            key = nt_pack4(d1) | (nt_pack4(d2) << 8) | (nt_pack4(d3) << 16) | (nt_pack4(d4 & 0xFFFFu) << 24);
            const uint32_t b = hash_u64(key, a.bucket_bits);
            uint32_t word;
            const uint64_t m = filter_mask(key, a.bucket_bits, word);
            if ((sd != PEP_SELF_NO_DELTA && a.debug != 11) || (a.filter[word] & m) == m) {      // (as in seed_match: a self stretch's keys are in the index; one request for both ends of the bucket)
                if (a.debug == 12) { e0 = a.start[b]; e1 = a.start[b + 1]; }
                else { uint32_t ends[2]; __builtin_memcpy(ends, a.start + b, 8); e0 = ends[0]; e1 = ends[1]; }
            }
        }
        const uint32_t tl = d0, tr = (d4 >> 16) | (d5 << 16);           // target bytes p - 4 .. p - 1 and p + 14 .. p + 17
        for (uint32_t e = e0; e < e1; e += STRIDE_ENTRY_TRIP) {
            uint64_t pair[STRIDE_ENTRY_TRIP];
            __builtin_memcpy(pair, a.entries + e, 8 * STRIDE_ENTRY_TRIP);
#pragma unroll
            for (int k2 = 0; k2 < STRIDE_ENTRY_TRIP; ++k2) {
                const uint64_t ent = pair[k2];
                if (e + k2 >= e1 || (ent >> POS_BITS) != key) continue;
                const uint32_t qpos = (uint32_t)(ent & POS_MASK);
                if ((int32_t)(p - qpos) == sd) {
                    // a gene against itself on diagonal 0 (self_prepare: the target repeats the query base for base, with the same length, and has settled the candidate):
                    // the bases next to the word are the same on both sides - or padding on both -, so the target's side alone says how many there are; counted, not emitted
                    const uint32_t vl = tl & 0xFCFCFCFCu, vr = tr & 0xFCFCFCFCu;
                    const int Ls = min(NS - 1, (int)(__clz((int)vl) >> 3)), Rs = vr ? min(NS - 1, (__ffs((int)vr) - 1) >> 3) : NS - 1;
                    if (Ls >= NS - 1 - Rs) n_hit += (uint32_t)(Ls - (NS - 1 - Rs) + 1);
                    continue;
                }
                uint32_t ql, qr;                                        // query bytes qpos - 4 .. qpos - 1 and qpos + 14 .. qpos + 17 (>= 16 bytes of padding around every sequence)
                __builtin_memcpy(&ql, a.q_res + qpos - 4, 4);
                __builtin_memcpy(&qr, a.q_res + qpos + NK, 4);
                const uint32_t ml = (ql ^ tl) | (tl & 0xFCFCFCFCu), mr = (qr ^ tr) | (tr & 0xFCFCFCFCu);
                const int L = min(NS - 1, (int)(__clz((int)ml) >> 3));               // matching bases in front of the word (__clz(0) = 32)
                const int R = mr ? min(NS - 1, (__ffs((int)mr) - 1) >> 3) : NS - 1;  // ... behind it
                const int k_lo = NS - 1 - R, k_hi = L;
                if (k_hi < k_lo) continue;
                const uint32_t nh = (uint32_t)(k_hi - k_lo + 1);
                n_hit += nh;
                const uint32_t idx = atomicAdd(&nbuf, nh);
                for (int k = k_hi; k >= k_lo; --k) {                    // increasing target position
                    const uint64_t hit = ((uint64_t)(qpos - k) << 32) | (uint64_t)(p - k);
                    const uint32_t at = idx + (uint32_t)(k_hi - k);
                    if (at < HIT_BUF) buf[at] = hit;
                    else {
                        const unsigned long long g = atomicAdd(a.hit_count, 1ull);
                        if (g < a.hit_cap) a.hits[g] = hit; else a.counters[2] = 1u;
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t cnt = nbuf;
        if (cnt > HIT_BUF / 2 || next_tile >= n_tiles) {
            const uint32_t n = min(cnt, (uint32_t)HIT_BUF);
            if (threadIdx.x == 0) gbase = n ? atomicAdd(a.hit_count, (unsigned long long)n) : 0ull;
            __syncthreads();
            const unsigned long long g = gbase;
            for (uint32_t i = threadIdx.x; i < n; i += 256) {
                if (g + i < a.hit_cap) a.hits[g + i] = buf[i]; else a.counters[2] = 1u;
            }
            __syncthreads();
            if (threadIdx.x == 0) nbuf = 0;
            __syncthreads();
        }
        tile = next_tile;
    }
    for (int d = 32; d > 0; d >>= 1) {
        n_seed += __shfl_down(n_seed, d, 64);
        n_hit += __shfl_down(n_hit, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&blk_stats[0], n_seed); atomicAdd(&blk_stats[1], n_hit); }
    __syncthreads();
    if (threadIdx.x < 2 && blk_stats[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)blk_stats[threadIdx.x]);
}

// Phase 2: raw seed hits -> runs of one candidate -> ungapped x-drop extensions -> candidate set, in ONE kernel (round 4; seed_runs + seed_extend
// before, with the runs written to and read back from global memory in between: 20 bytes per run, 2 GB per shape at 50 000 genes).
//
// Runs.  Neighbouring hits of the buffer usually come from neighbouring positions of one diagonal, i.e. they nominate the same candidate
// (q, t, diagonal bin; q / t / the diagonal from the block -> sequence maps): every run of equal candidate keys inside a wavefront's 64 hits
// is ONE work item.  A candidate needs one hit whose extension reaches the threshold: the hits of a run are tried in order until one passes
// (true homologues pass at the first or second, chance runs are one or two hits long); the first passer inserts the key.  The set is
// order-independent, so the result does not depend on scheduling.
//
// Execution.  Every wavefront works for itself - no block-level barrier anywhere.  It turns RUN_TRIP x 64 hits into runs, kept in its own
// strip of LDS, and then advances extensions in ROUNDS: every lane holds one extension in flight and a round moves all of them on by one
// block of sixteen residues, whichever side and block each is in (the piece of the two windows is fetched for it: 16 bytes per sequence; the
// first block is stage 1).  A lane whose extension is decided takes the next run of the strip in the same round; when the strip is empty the
// wavefront makes the next one while the extensions still in flight keep their state.  Before, a thread walked its run through 96 unrolled,
// predicated residues: as long as ONE lane of a wavefront was extending all of it was issued - 1 100 scalar and 650 vector instructions
// per wavefront and slice (SQ_INSTS_SALU 44.6 M against SQ_INSTS_VALU 26.2 M per launch at 10 k genes), most of them for the one or two
// lanes whose extension went beyond its first sixteen residues.
struct XDrop { int s, best, live, pass; };      // live / pass: 0 or 1
// one residue pair, for every lane and without a branch: the state of a lane that is not extending (on = 0) does not change.
//   s = running score, best = its maximum so far, base = what the other side of the seed has secured (0 on the right side)
//   pass: base + best reached the threshold;  an extension ends when it passes or falls more than xdrop below its best
__device__ __forceinline__ void xdrop_step(XDrop &x, int sc, int on, int base, int thr, int xdrop)
{
    const int act = x.live & on;
    const int s2 = x.s + sc;
    const int up = s2 > x.best ? 1 : 0;
    const int nb = up ? s2 : x.best;
    const int p = up & (base + nb >= thr ? 1 : 0);
    const int d = (up ^ 1) & (x.best - s2 > xdrop ? 1 : 0);
    x.s = act ? s2 : x.s;
    x.best = act ? nb : x.best;
    x.pass |= act & p;
    x.live &= (act & (p | d)) ^ 1;
}

constexpr int RUN_TRIP = 4;                     // rounds of 64 hits a wavefront turns into runs at a time (their look-up chains overlap)
constexpr int RUN_STRIP = 64 * RUN_TRIP;        // runs a wavefront's strip holds (a run per hit at most)
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ __launch_bounds__(256) void seed_runs_extend(JoinArgs a)
{
    __shared__ int8_t sub[1024];
    __shared__ uint32_t blk_pass;
    __shared__ uint64_t s_first[4][RUN_STRIP], s_key[4][RUN_STRIP];
    __shared__ uint32_t s_len[4][RUN_STRIP];
    reinterpret_cast<uint32_t *>(sub)[threadIdx.x] = reinterpret_cast<const uint32_t *>(a.sub)[threadIdx.x];
    if (threadIdx.x == 0) blk_pass = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *q_first = s_first[wave], *q_key = s_key[wave];
    uint32_t *q_len = s_len[wave];
    unsigned long long n_hits = *a.hit_count;
    if (n_hits > a.hit_cap) n_hits = a.hit_cap;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * RUN_TRIP;
    uint64_t next_hits = ((uint64_t)blockIdx.x * 4 + wave) * 64 * RUN_TRIP;      // this wavefront's next slice of the hit buffer
    uint32_t q_n = 0, q_take = 0;                                                // the strip: runs made, runs taken (wave-uniform)
    uint32_t n_pass = 0;
    // the extension a lane has in flight
    int busy = 0, side = 0, k = 0, br = 0;
    XDrop X = {0, 0, 0, 0};
    uint64_t ck = 0, first = 0;
    uint32_t len = 0, h = 0, qp = 0, tp = 0;
    for (;;) {
        // ---- the strip is empty: the next RUN_TRIP x 64 hits become runs
        if (q_take == q_n && next_hits < n_hits) {
            wave_sync_lds();                                                     // (every lane has read what it took from the strip)
            q_n = q_take = 0;
            uint64_t hh[RUN_TRIP], key[RUN_TRIP], hit[RUN_TRIP];
            bool valid[RUN_TRIP];
#pragma unroll
            for (int u = 0; u < RUN_TRIP; ++u) {
                hh[u] = next_hits + (uint64_t)u * 64 + lane;
                valid[u] = hh[u] < n_hits;
                hit[u] = valid[u] ? a.hits[hh[u]] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < RUN_TRIP; ++u) {
                key[u] = ~0ull;
                if (valid[u]) {
                    const uint32_t hq = (uint32_t)(hit[u] >> 32), p = (uint32_t)hit[u];
                    const uint2 tb = a.t_blk2seq[p >> 5], qb = a.q_blk2seq[hq >> 5];          // one look-up per side (sequence and its start together: not two dependent ones)
                    const int32_t diag = (int32_t)(p - tb.y) - (int32_t)(hq - qb.y);
                    const uint32_t bin = (uint32_t)(diag + (1 << 23)) >> 6;
                    key[u] = ((uint64_t)qb.x << 43) | ((uint64_t)tb.x << 18) | (uint64_t)bin;
                }
            }
            bool leader[RUN_TRIP];
            int rlen[RUN_TRIP];
            uint64_t seen[RUN_TRIP];
#pragma unroll
            for (int u = 0; u < RUN_TRIP; ++u) {
                const uint32_t lo = (uint32_t)key[u], hi = (uint32_t)(key[u] >> 32);
                const uint32_t lo_prev = __shfl_up(lo, 1, 64), hi_prev = __shfl_up(hi, 1, 64);     // unconditional: every lane must take part in the shuffles
                leader[u] = valid[u] && !(a.debug != 9 && lane > 0 && lo_prev == lo && hi_prev == hi);
                const unsigned long long leaders = __ballot(leader[u]), valid_m = __ballot(valid[u]);
                const unsigned long long above = lane == 63 ? 0ull : (leaders & (~0ull << (lane + 1)));          // the next run starts at its lowest set bit
                rlen[u] = (above ? __builtin_ctzll(above) : __popcll(valid_m)) - lane;                            // valid lanes are a prefix
            }
            // a run of ONE hit is almost always a chance hit of the reduced alphabet whose candidate is in no set: it goes straight to the
            // extension (which decides) instead of paying a scattered probe of the set; longer runs - homologous diagonals - are dropped HERE
            // when their candidate is established already (an earlier run, shape or launch).  The first probes of the whole trip are in flight
            // together; only a slot that holds another key sends a lane down the probe sequence.
#pragma unroll
            for (int u = 0; u < RUN_TRIP; ++u) seen[u] = (leader[u] && rlen[u] > 1) ? a.table[hash_u64(key[u], a.table_bits)] : EMPTY;
#pragma unroll
            for (int u = 0; u < RUN_TRIP; ++u) {
                bool queue = leader[u];
                if (leader[u] && rlen[u] > 1 && (seen[u] == key[u] || (seen[u] != EMPTY && set_contains(a, key[u])))) { queue = false; ++n_pass; }
                const unsigned long long queued = __ballot(queue);
                if (queue) {
                    const uint32_t slot = q_n + (uint32_t)__popcll(queued & ((1ull << lane) - 1ull));
                    // (a run of one carries its hit itself - no look-up into the hit buffer later; longer runs the index of their first hit)
                    q_first[slot] = rlen[u] == 1 ? hit[u] : hh[u]; q_len[slot] = (uint32_t)rlen[u]; q_key[slot] = key[u];
                }
                q_n += (uint32_t)__popcll(queued);
            }
            next_hits += stride;
            wave_sync_lds();
        }
        // ---- idle lanes take the next runs of the strip
        const unsigned long long idle = __ballot(!busy);
        if (idle && q_take < q_n) {
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull)), r = q_take + rank;
            if (!busy && r < q_n) {
                ck = q_key[r]; first = q_first[r]; len = q_len[r];
                if (a.ungapped_min <= 0) { ++n_pass; set_insert(a, ck); }
                else {
                    const uint64_t hit = len == 1 ? first : a.hits[first];
                    qp = (uint32_t)(hit >> 32); tp = (uint32_t)hit; h = 0;
                    busy = 1; side = 0; k = 0; X = XDrop{0, 0, 1, 0};
                }
            }
            q_take = min(q_n, q_take + (uint32_t)__popcll(idle));
        }
        if (!__ballot(busy)) {
            if (q_take == q_n && next_hits >= n_hits) break;                     // nothing in flight, nothing in the strip, no hits left
            continue;
        }
        // ---- one block of XB residues for every extension in flight: right side block k = residues XB k .. XB k + XB - 1 from the seed start,
        // left side block k = residues XB k + 1 .. XB k + XB before it (read from the seed outwards).  XB = 16: the first block IS stage 1
        // (pep_search_params.stage1_min; oracle: ungapped_score).  Padding bytes (>= 16 around every sequence) score -64, which ends an
        // extension exactly where the sequence ends (x-drop < 64): no bounds are needed.
        constexpr int XB = 16;
        uint32_t qw[XB / 4], tw[XB / 4];
#pragma unroll
        for (int w = 0; w < XB / 4; ++w) qw[w] = tw[w] = 0;
        if (busy) {
            const int off = side ? -XB * (k + 1) : XB * k;
            __builtin_memcpy(qw, a.q_res + (int64_t)qp + off, XB);
            __builtin_memcpy(tw, a.t_res + (int64_t)tp + off, XB);
            if (side) {                                                        // left side: nearest residue first
                uint32_t q[XB / 4], t[XB / 4];
#pragma unroll
                for (int w = 0; w < XB / 4; ++w) { q[w] = __builtin_bswap32(qw[XB / 4 - 1 - w]); t[w] = __builtin_bswap32(tw[XB / 4 - 1 - w]); }
#pragma unroll
                for (int w = 0; w < XB / 4; ++w) { qw[w] = q[w]; tw[w] = t[w]; }
            }
        }
        const int r0 = side ? XB * k + 1 : XB * k, lim = side ? a.ext_left + 1 : a.ext_right, base_score = side ? br : 0;
#pragma unroll
        for (int j = 0; j < XB; ++j) {
            const int qc = (int)((qw[j >> 2] >> ((j & 3) * 8)) & 31u), tc = (int)((tw[j >> 2] >> ((j & 3) * 8)) & 31u);
            xdrop_step(X, sub[qc * 32 + tc], busy & (r0 + j < lim ? 1 : 0), base_score, a.ungapped_min, a.xdrop);
        }
        // ---- where the extension stands after the block
        if (busy) {
            ++k;
            bool next_hit = false;
            if (X.pass) { ++n_pass; set_insert(a, ck); busy = 0; }
            else if (side == 0) {
                // stage 1: after sixteen residues to the right (or wherever the extension ended before) it must have reached stage1_min
                const bool over = !X.live || XB * k >= a.ext_right;
                if (k == 1 && X.best < a.stage1_min) next_hit = true;
                else if (over) { side = 1; k = 0; br = X.best; X = XDrop{0, 0, 1, 0}; }
            } else if (!X.live || XB * k + 1 > a.ext_left) next_hit = true;
            if (next_hit) {
                if (++h < len) {
                    const uint64_t hit = a.hits[first + h];
                    qp = (uint32_t)(hit >> 32); tp = (uint32_t)hit;
                    side = 0; k = 0; X = XDrop{0, 0, 1, 0};
                } else busy = 0;
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) n_pass += __shfl_down(n_pass, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&blk_pass, n_pass);
    __syncthreads();
    if (threadIdx.x == 0 && blk_pass) atomicAdd(&a.stats[2], (unsigned long long)blk_pass);
}

// Self-search (PEPPAN's hot call is one: -r CL -q CL): one wavefront per query g, in front of the matchers.
//  1. Is the packed sequence that frame 1 of reference sequence g starts with (first_t[g], left by K1) query g over again - the same residues from its first
//     position on, at least the query's length?  Decided from the data; a target that is not is left to the stream as it is.
//  2. If so, the 32-byte blocks of the target layout inside [ts, ts + ql) get delta = ts - qs: seed_match counts and DROPS the hits of those blocks whose query
//     position is the target position - delta - the gene against itself on diagonal 0, 57 % of the raw hits of the 10 000-gene search.  (The block that holds
//     ts begins in the padding in front of the target; a block that reaches beyond ts + ql is left out: behind the query's end lies another query.)
//  3. Their candidate is settled here.  The candidate set is a SET, and (q, t, bin of diagonal 0) is in it iff ONE seed hit of that bin passes the ungapped
//     pre-filter (oracle: find_candidates / ungapped_score).  The hits on diagonal 0 follow from the sequence alone: every position of the query that starts a
//     seed of one of the shapes is a hit against the target that repeats it (same residues, same key, in the index by construction), and its extension is the
//     oracle's loop over the packed residues (padding scores -64 and ends an extension where a sequence ends: x-drop < 64).  64 positions per round until one
//     passes - the first round, as a rule; an extension stops as soon as its right side alone has reached the threshold (it can only grow).  Hits of the same
//     pair that the matcher does not drop (the last, partial block; other diagonals of the bin) take the usual way; inserting a key twice is harmless.
__global__ __launch_bounds__(256) void self_prepare(SeedShapeSet shs, int n_shapes, JoinArgs a, const uint32_t *__restrict__ first_t, uint32_t n_first,
                                                    const uint32_t *__restrict__ q_len, const uint32_t *__restrict__ t_len, int32_t *__restrict__ delta, int exact_len)
{
    __shared__ int8_t sub[1024];
    reinterpret_cast<uint32_t *>(sub)[threadIdx.x] = reinterpret_cast<const uint32_t *>(a.sub)[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= a.nq || g >= n_first) return;
    const uint32_t t = first_t[g];
    if (t >= a.nt) return;                                      // (PEP_SELF_NONE included)
    const uint32_t ql = q_len[g], tl = t_len[t];
    if (ql == 0u || tl < ql || (exact_len && tl != ql)) return;
    const uint32_t qs = a.q_off[g], ts = a.t_off[t];             // (16-aligned starts: the 8-byte reads below are aligned)
    bool same = true;
    for (uint32_t x = (uint32_t)lane * 8u; x < ql && same; x += 512u) {
        if (x + 8u <= ql) same = *reinterpret_cast<const uint64_t *>(a.q_res + qs + x) == *reinterpret_cast<const uint64_t *>(a.t_res + ts + x);
        else for (uint32_t y = x; y < ql; ++y) same = same && a.q_res[qs + y] == a.t_res[ts + y];
    }
    if (!__all(same)) return;
    for (uint32_t b = (ts >> 5) + (uint32_t)lane, b1 = (ts + ql) >> 5; b < b1; b += 64u) delta[b] = (int32_t)(ts - qs);
    if (a.ungapped_min <= 0) return;                            // (no pre-filter: every hit nominates; the caller does not drop hits then)
    constexpr int STAGE1_LEN = 16;
    const int thr = a.ungapped_min, early = max(a.ungapped_min, a.stage1_min);
    for (int s = 0; s < n_shapes; ++s) {
        const SeedShape &sh = shs.s[s];
        for (uint32_t x0 = 0; x0 < ql; x0 += 64) {
            const uint32_t x = x0 + (uint32_t)lane;
            bool pass = false;
            if (x < ql) {
                const uint8_t *q = a.q_res + qs + x, *tt = a.t_res + ts + x;
                bool seeds = true;
                for (int i = 0; i < sh.weight; ++i) seeds = seeds && reduce_letter(sh, q[sh.offs[i]]) != 15u;
                if (seeds) {
                    int sc = 0, br = 0, bl = 0, k = 0;
                    bool dead = false, over = false;
                    // the first sixteen residues of the right side from four 8-byte reads issued together (a loop of byte reads that may end at every step is a chain
                    // of dependent round trips: 30 us for the 10 000 genes); the rest - rarely needed - byte by byte
                    uint64_t qw[2], tw[2];
                    __builtin_memcpy(qw, q, 16);
                    __builtin_memcpy(tw, tt, 16);
#pragma unroll
                    for (int j = 0; j < STAGE1_LEN; ++j) {
                        const int qc = (int)((qw[j >> 3] >> ((j & 7) * 8)) & 31u), tc = (int)((tw[j >> 3] >> ((j & 7) * 8)) & 31u);
                        if (!over && j < a.ext_right) {
                            sc += sub[qc * 32 + tc];
                            if (sc > br) { br = sc; over = br >= early; }                // passed whatever follows: br never falls, stage 1 is met
                            else over = br - sc > a.xdrop;
                            k = j + 1;
                        }
                    }
                    if (!over) for (k = STAGE1_LEN; k < a.ext_right; ++k) {
                        if (k == STAGE1_LEN && br < a.stage1_min) { dead = true; break; }
                        sc += sub[(q[k] & 31) * 32 + (tt[k] & 31)];
                        if (sc > br) { br = sc; if (br >= early) break; }
                        else if (br - sc > a.xdrop) break;
                    }
                    if (!dead && br >= a.stage1_min) {
                        if (br < thr) {
                            sc = 0;
                            for (k = 1; k <= a.ext_left; ++k) {
                                sc += sub[(q[-k] & 31) * 32 + (tt[-k] & 31)];
                                if (sc > bl) { bl = sc; if (br + bl >= thr) break; }
                                else if (bl - sc > a.xdrop) break;
                            }
                        }
                        pass = br + bl >= thr;
                    }
                }
            }
            if (__ballot(pass)) {
                // (not counted in stats[2], the passed-hits statistic: ten thousand wavefronts adding to one word take 110 us - the word sustains ~90 atomics/us)
                if (lane == 0) set_insert(a, ((uint64_t)g << 43) | ((uint64_t)t << 18) | (uint64_t)((1u << 23) >> 6));      // diagonal 0: bin (0 + 2^23) >> 6
                return;
            }
        }
    }
}

// hash set -> dense list (arbitrary order; sorted afterwards).  One global atomic per block and per COMPACT_ROUNDS x 256 slots:
// the single counter word sustains ~90 atomics/us, so 4096 blocks with one atomic each spent 45 us waiting for it.
constexpr int COMPACT_ROUNDS = 16;
// The keys leave in the dense form q | t | bin - bin_min (only as many bits per field as this search needs: the radix sort then runs over
// ~36 instead of 64 bits; its last pass restores q:21 | t:25 | bin:18), and every slot that held a key is EMPTY again afterwards: the
// next search finds the set clean instead of filling 8 MB in front of its first kernel.
__global__ __launch_bounds__(256) void set_compact(uint64_t *__restrict__ table, uint64_t cap, uint64_t *__restrict__ list, uint32_t list_cap,
                                                   uint32_t *__restrict__ counters, int tb, int bb, uint32_t bin_min, uint32_t *__restrict__ top_hist, int top_shift)
{
    __shared__ uint32_t wave_cnt[4], blk_base;
    __shared__ uint32_t top[1 << PEP_SORT_TOP_BITS];          // keys of this block per top digit of the dense key (top_shift < 0: not wanted)
    if (top_shift >= 0) for (int x = threadIdx.x; x < (1 << PEP_SORT_TOP_BITS); x += 256) top[x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t k[COMPACT_ROUNDS];
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < COMPACT_ROUNDS; ++r) {
        const uint64_t i = ((uint64_t)blockIdx.x * COMPACT_ROUNDS + r) * 256 + threadIdx.x;
        k[r] = i < cap ? table[i] : EMPTY;
        if (k[r] != EMPTY) { ++mine; table[i] = EMPTY; }
    }
    // exclusive prefix of `mine` over the block
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        blk_base = tot ? atomicAdd(&counters[0], tot) : 0u;
    }
    __syncthreads();
    uint32_t idx = blk_base + incl - mine;
    for (int w = 0; w < wave; ++w) idx += wave_cnt[w];
#pragma unroll
    for (int r = 0; r < COMPACT_ROUNDS; ++r)
        if (k[r] != EMPTY) {
            const uint64_t q = k[r] >> 43, t = (k[r] >> 18) & ((1u << 25) - 1), bin = (k[r] & ((1u << 18) - 1)) - bin_min;
            const uint64_t dense = (q << (tb + bb)) | (t << bb) | bin;
            if (idx < list_cap) list[idx] = dense; else counters[1] = 1u;
            if (top_shift >= 0) atomicAdd(&top[(uint32_t)(dense >> top_shift) & ((1u << PEP_SORT_TOP_BITS) - 1)], 1u);
            ++idx;
        }
    if (top_shift >= 0) {
        __syncthreads();
        for (int x = threadIdx.x; x < (1 << PEP_SORT_TOP_BITS); x += 256) if (top[x]) atomicAdd(&top_hist[x], top[x]);
    }
}

int ilog2_ceil(uint64_t x)
{
    int b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

}  // namespace

// the plain 32 x 32 substitution table of the current parameters in device memory (ctx->d_params): the ungapped extension of the seed
// stage and the gapless shortcut of the traceback stage read it
int pep_upload_sub_table(pep_ctx *ctx)
{
    const pep_search_params &P = ctx->params;
    PEP_TRY(dev_reserve(ctx, ctx->d_params, 1024));
    if (!ctx->d_params_valid || memcmp(ctx->d_params_host, P.sub, 1024) != 0) {
        // (a copy out of pageable memory makes the host wait for the stream: only when the table really changed)
        PEP_HIP(ctx, hipMemcpyAsync(ctx->d_params.p, P.sub, 1024, hipMemcpyHostToDevice, ctx->stream));
        memcpy(ctx->d_params_host, P.sub, 1024);
        ctx->d_params_valid = true;
    }
    return PEP_OK;
}

// workspace slots used here: ws[0] cnt, ws[1] start, ws[2] entries, ws[3] table, ws[4] list, ws[5] list tmp (sort),
// ws[6] counters+stats, ws[7] scan scratch, ws[8] raw seed hits, ws[10..12] runs of hits (the sort histogram reuses ws[0])
// need_targets (optional): the target set may still be on its way (pep_search queues K1 for both sides and waits for the query side only);
// the hook is called once, right before the first use of anything about the targets - after the first shape's index build has been queued,
// so that the GPU goes from K1 into the index kernels while the host picks up the target side's summary.
int pep_find_candidates(pep_ctx *ctx, uint64_t **d_cands, uint64_t *n_cands, int (*before_sync)(pep_ctx *), int (*need_targets)(pep_ctx *))
{
    const pep_search_params &P = ctx->params;
    SeqSet &Q = ctx->q, &T = ctx->t;
    if (Q.total > PEP_MAX_RESIDUES) return pep_fail(ctx, PEP_ERR_LIMIT, "more than 2^29 packed residues on one side");
    *n_cands = 0;
    *d_cands = nullptr;
    ctx->stats.query_seeds = ctx->stats.target_seeds = ctx->stats.seed_hits = 0;
    if (Q.n == 0) {
        if (need_targets) PEP_TRY(need_targets(ctx));
        if (before_sync) PEP_TRY(before_sync(ctx));
        ctx->upload.n_words = 0;                               // (nothing to align: the thresholds are not needed)
        return PEP_OK;
    }

    // two buckets per query position, except that up to 40 M positions stay at 2^25 buckets (the average coarse bucket then holds 4 900
    // of the 5 632 entries a slab takes; 100k genes x 100k genes: 103 -> 94 ms per pass): that is the largest index the partition
    // build handles (2^13 coarse x 2^12 fine buckets), which is worth more than the last halving of the load (the filter in front of
    // the index keeps most foreign keys away from the buckets anyway)
    int bucket_bits = std::max(10, std::min(28, ilog2_ceil(2 * Q.total)));
    if (bucket_bits > 25 && Q.total <= 40000000ull && P.reserved[2] == 0) bucket_bits = 25;
    const uint64_t n_buckets = 1ull << bucket_bits;
    // every shape has an index of its own (start[], entries, filter): the partition build makes all of them in one launch per kernel
    // (blockIdx.y = shape), and the matcher of shape s then reads copy s.  Shapes of different weights (no caller has them) share copy 0, one after the other.
    bool same_weight = true;
    for (int s = 1; s < P.n_shapes && s < 4; ++s) same_weight = same_weight && P.weight[s] == P.weight[0];
    const int n_idx = same_weight ? std::max(1, std::min<int>(P.n_shapes, 4)) : 1;
    const uint64_t start_stride = (n_buckets + 2 + 15) & ~15ull, entries_stride = (Q.total + 4 + 7) & ~7ull, filter_stride = ((n_buckets >> FILTER_SHIFT) + 2 + 7) & ~7ull;
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], (n_buckets + 1) * sizeof(uint32_t)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[1], n_idx * start_stride * sizeof(uint32_t)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[2], n_idx * entries_stride * sizeof(uint64_t)));
    PEP_TRY(dev_reserve(ctx, ctx->d_zero, PEP_ZERO_TOTAL));
    PEP_TRY(dev_reserve(ctx, ctx->ws[9], n_idx * filter_stride * 8));
    unsigned long long *filter0 = ctx->ws[9].as<unsigned long long>();
    uint32_t *cnt = ctx->ws[0].as<uint32_t>(), *start0 = ctx->ws[1].as<uint32_t>();
    uint64_t *entries0 = ctx->ws[2].as<uint64_t>();
    char *zero = ctx->d_zero.as<char>();
    uint32_t *counters = reinterpret_cast<uint32_t *>(zero + PEP_ZERO_SEED);
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(counters + 4);
    PEP_TRY(pep_upload_sub_table(ctx));

    // candidate set: start near 64 slots per query (chance hits grow with |Q| x |T|), grow x4 on overflow
    int table_bits = std::max(20, std::min(28, ilog2_ceil(64ull * Q.n)));
    uint64_t hit_cap = 0;                          // raw seed hits: 2 x target bytes to start with (sized below, once the targets are known)
    if (P.n_shapes > 4) return pep_fail(ctx, PEP_ERR_ARG, "more than 4 seed shapes");
    // query index build: by partition when the bucket count splits into <= 2^13 coarse x <= 2^12 fine buckets (reserved[2] != 0 forces the
    // count -> scan -> fill build, as does a coarse bucket that overflows LDS)
    const int fine_bits = std::min(12, bucket_bits - 8);
    // the nucleotide tool (exact 17-mers over the four bases, one shape): look-up words at a stride (seed_match_stride); reserved[0] = 8 keeps the plain matcher
    bool stride_lookup = P.n_shapes == 1 && P.base == 4 && P.weight[0] == NW && (P.reserved[0] == 0 || P.reserved[0] == 11 || P.reserved[0] == 12);
    for (int i = 0; i < NW && stride_lookup; ++i) stride_lookup = P.offs[0][i] == i;
    for (int c = 0; c < 32 && stride_lookup; ++c) stride_lookup = P.reduce[c] == (c < 4 ? c : 0xFF);
    bool use_partition = P.reserved[2] == 0 && bucket_bits >= 16 && bucket_bits - fine_bits <= 13;
    for (int attempt = 0; attempt < 8; ++attempt) {
        const uint64_t cap = 1ull << table_bits;
        const uint32_t list_cap = (uint32_t)(cap >> 1);
        // the candidate set lives in a buffer of its own: set_compact hands every slot back EMPTY, so only a new (or larger, or abandoned)
        // table is filled here
        if (cap * sizeof(uint64_t) > ctx->d_set.cap) ctx->set_clean_slots = 0;
        PEP_TRY(dev_reserve(ctx, ctx->d_set, cap * sizeof(uint64_t)));
        if (ctx->set_clean_slots < cap) PEP_HIP(ctx, hipMemsetAsync(ctx->d_set.p, 0xFF, cap * sizeof(uint64_t), ctx->stream));
        ctx->set_clean_slots = 0;                               // in use until set_compact below has been queued
        PEP_TRY(dev_reserve(ctx, ctx->ws[4], (uint64_t)list_cap * sizeof(uint64_t)));
        PEP_TRY(dev_reserve(ctx, ctx->ws[5], (uint64_t)list_cap * sizeof(uint64_t)));
        // ONE fill for every small counter block of the search (seed stage, candidate sort, both Smith-Waterman passes, selection)
        // (not even that when the previous search's last kernel has left all of them cleared and nothing has touched them since)
        bool all_ok = ctx->zero_clean;
        for (bool f : ctx->zero_ok) all_ok = all_ok && f;
        if (!all_ok) PEP_HIP(ctx, hipMemsetAsync(zero, 0, PEP_ZERO_TOTAL, ctx->stream));
        ctx->zero_clean = false;
        for (bool &f : ctx->zero_ok) f = true;
        uint64_t q_seeds = 0;
        int self_on = 0;
        auto make_shape = [&](int s) {
            SeedShape sh;
            sh.weight = stride_lookup ? NK : P.weight[s];      // (the look-up word of seed_match_stride: a prefix of the contiguous NW-mer)
            sh.base = P.base;
            for (int i = 0; i < 32; ++i) sh.offs[i] = P.offs[s][i];
            for (int w = 0; w < 4; ++w) sh.red4[w] = 0;
            for (int c = 0; c < 32; ++c) {
                const uint32_t g = P.reduce[c] == 0xFF ? 15u : (uint32_t)P.reduce[c];
                sh.red4[c >> 3] |= g << ((c & 7) * 4);
            }
            sh.h1 = (sh.weight + 1) / 2;
            sh.pow_h1 = 1;
            for (int i = 0; i < sh.h1; ++i) sh.pow_h1 *= (uint32_t)sh.base;
            for (int i = 0; i < 32; ++i) sh.pw[i] = 0;
            for (int i = 0, pl = 1, ph = 1; i < sh.weight; ++i) {
                if (i < sh.h1) { sh.pw[i] = (uint32_t)pl; pl *= sh.base; } else { sh.pw[i] = (uint32_t)ph; ph *= sh.base; }
            }
            return sh;
        };
        const bool fused_build = use_partition && n_idx > 1;                   // all shapes' indices by the first trip through the loop
        for (int s = 0; s < P.n_shapes; ++s) {
            const SeedShape sh = make_shape(s);
            const int copy = n_idx > 1 ? s : 0;
            uint32_t *start = start0 + copy * start_stride;
            uint64_t *entries = entries0 + copy * entries_stride;
            unsigned long long *filter = filter0 + copy * filter_stride;
            const unsigned qb = (unsigned)ceil_div(Q.total, 256);
#define PEP_SEED_DISPATCH(KERNEL, GRID, ...)                                                                          \
    do {                                                                                                              \
        if (sh.weight == 10) hipLaunchKernelGGL(KERNEL<10>, GRID, dim3(256), 0, ctx->stream, __VA_ARGS__);            \
        else if (sh.weight == 17) hipLaunchKernelGGL(KERNEL<17>, GRID, dim3(256), 0, ctx->stream, __VA_ARGS__);       \
        else if (sh.weight == 14) hipLaunchKernelGGL(KERNEL<14>, GRID, dim3(256), 0, ctx->stream, __VA_ARGS__);       \
        else hipLaunchKernelGGL(KERNEL<0>, GRID, dim3(256), 0, ctx->stream, __VA_ARGS__);                             \
    } while (0)
#define PEP_SEED_DISPATCH_LDS(KERNEL, GRID, LDS, ...)                                                                 \
    do {                                                                                                              \
        if (sh.weight == 10) hipLaunchKernelGGL(KERNEL<10>, GRID, dim3(256), LDS, ctx->stream, __VA_ARGS__);          \
        else if (sh.weight == 17) hipLaunchKernelGGL(KERNEL<17>, GRID, dim3(256), LDS, ctx->stream, __VA_ARGS__);     \
        else if (sh.weight == 14) hipLaunchKernelGGL(KERNEL<14>, GRID, dim3(256), LDS, ctx->stream, __VA_ARGS__);     \
        else hipLaunchKernelGGL(KERNEL<0>, GRID, dim3(256), LDS, ctx->stream, __VA_ARGS__);                           \
    } while (0)
            if (use_partition) {
                // partition build: entries into per-coarse-bucket slabs (one pass) -> per-bucket LDS sort
                const int tiles = (int)std::max<uint64_t>(PART_TILES, ceil_div(Q.total, (uint64_t)TILE * 2048));
                const unsigned pb = (unsigned)ceil_div(Q.total, (uint64_t)tiles * TILE);
                const uint32_t n_coarse = 1u << (bucket_bits - fine_bits);
                const size_t lds = (size_t)n_coarse * 4 + TILE + TILE_HALO;
                const uint64_t part_stride = (uint64_t)n_coarse * PART_CAP;
                PEP_TRY(dev_reserve(ctx, ctx->ws[13], (fused_build ? n_idx : 1) * part_stride * sizeof(uint64_t)));
                uint64_t *part = ctx->ws[13].as<uint64_t>();
                uint32_t *coarse = reinterpret_cast<uint32_t *>(zero + PEP_ZERO_COARSE) + (size_t)s * 8192;       // (cleared by the search's one fill)
                if (!fused_build || s == 0) {
                    SeedShapeSet shs;
                    const unsigned ny = fused_build ? (unsigned)n_idx : 1u;
                    for (unsigned y = 0; y < 4; ++y) shs.s[y] = make_shape(fused_build ? (int)std::min<unsigned>(y, ny - 1) : s);
                    PEP_SEED_DISPATCH_LDS(idx_slab, dim3(pb, ny), lds, shs, Q.res.as<const uint8_t>(), Q.total, bucket_bits, fine_bits, coarse, part, part_stride, tiles, counters);
                    hipLaunchKernelGGL(idx_finish, dim3(n_coarse, ny), dim3(256), 0, ctx->stream, (const uint64_t *)part, (const uint32_t *)coarse, bucket_bits, fine_bits,
                                       start, entries, filter, counters, counters + 10 + s, part_stride, start_stride, entries_stride, filter_stride);
                }
            } else {
                PEP_HIP(ctx, hipMemsetAsync(cnt, 0, (n_buckets + 1) * sizeof(uint32_t), ctx->stream));
                PEP_SEED_DISPATCH(seed_count, dim3(qb), sh, Q.res.as<const uint8_t>(), Q.total, cnt, bucket_bits);
                PEP_TRY(pep_scan_u32(ctx, cnt, start, n_buckets, ctx->ws[7]));
                PEP_HIP(ctx, hipMemsetAsync(cnt, 0, (n_buckets + 1) * sizeof(uint32_t), ctx->stream));
                PEP_SEED_DISPATCH(seed_fill, dim3(qb), sh, Q.res.as<const uint8_t>(), Q.total, (const uint32_t *)start, cnt, entries, bucket_bits);
                PEP_HIP(ctx, hipMemsetAsync(filter, 0, (n_buckets >> FILTER_SHIFT) * 8, ctx->stream));
                hipLaunchKernelGGL(filter_fill, dim3(2048), dim3(256), 0, ctx->stream, (const uint64_t *)entries, (const uint32_t *)(start + n_buckets), bucket_bits, filter, counters + 10 + s);
            }
            if (s == 0) {
                // everything about the targets from here on
                if (need_targets) { PEP_TRY(need_targets(ctx)); need_targets = nullptr; }
                if (T.total > PEP_MAX_RESIDUES) return pep_fail(ctx, PEP_ERR_LIMIT, "more than 2^29 packed residues on one side");
                if (T.n == 0) {                                                          // (the index kernels queued so far are harmless)
                    if (before_sync) PEP_TRY(before_sync(ctx));
                    ctx->upload.n_words = 0;
                    return PEP_OK;
                }
                if (hit_cap == 0) hit_cap = std::max<uint64_t>(1ull << 22, 2 * T.total);
                PEP_TRY(dev_reserve(ctx, ctx->ws[8], hit_cap * sizeof(uint64_t)));
            }
            const unsigned tb = (unsigned)ceil_div(T.total, 256);
            JoinArgs a;
            a.t_res = T.res.as<const uint8_t>(); a.t_total = T.total; a.t_off = T.off.as<const uint32_t>(); a.nt = T.n;
            a.q_off = Q.off.as<const uint32_t>(); a.nq = Q.n; a.q_blk2seq = Q.blk2seq.as<const uint2>(); a.t_blk2seq = T.blk2seq.as<const uint2>(); a.start = start; a.entries = entries; a.filter = filter; a.bucket_bits = bucket_bits;
            a.table = ctx->d_set.as<uint64_t>(); a.table_bits = table_bits;
            a.counters = counters; a.stats = stats;
            a.q_res = Q.res.as<const uint8_t>(); a.sub = ctx->d_params.as<const int8_t>();
            a.debug = P.reserved[0]; a.ungapped_min = P.ungapped_min; a.stage1_min = P.stage1_min; a.xdrop = P.xdrop; a.ext_right = P.ext_right; a.ext_left = P.ext_left;
            a.self_delta = nullptr;
            if (s == 0 && P.ungapped_min > 0 && (P.reserved[0] == 0 || P.reserved[0] == 11 || P.reserved[0] == 12)) PEP_TRY(pep_self_map(ctx, &self_on));     // (reserved[0] = 10 / 8: the plain stream, for comparison)
            if (self_on) a.self_delta = ctx->d_self_delta.as<const int32_t>();
            unsigned long long *hit_count = reinterpret_cast<unsigned long long *>(zero + PEP_ZERO_SHAPE) + 2 * s;      // per shape, cleared by the one fill
            a.hits = ctx->ws[8].as<uint64_t>(); a.hit_count = hit_count; a.hit_cap = hit_cap;
            if (self_on && s == 0) {
                // which targets repeat a query, the blocks whose diagonal-0 self hits the matchers drop, and those genes' own candidates: one launch, all shapes
                SeedShapeSet shs;
                for (int y = 0; y < 4; ++y) { shs.s[y] = make_shape(std::min(y, P.n_shapes - 1)); shs.s[y].weight = P.weight[std::min(y, P.n_shapes - 1)]; }      // (the seed itself, not the stride matcher's look-up word)
                hipLaunchKernelGGL(self_prepare, dim3((unsigned)ceil_div(Q.n, 4)), dim3(256), 0, ctx->stream, shs, P.n_shapes, a, ctx->self_first, ctx->self_first_cnt,
                                   Q.len.as<const uint32_t>(), T.len.as<const uint32_t>(), ctx->d_self_delta.as<int32_t>(), ctx->self_exact_len);
                PEP_HIP(ctx, hipGetLastError());
            }
            pep_timer_begin(ctx, TM_MATCH0 + s);
            a.tile_ctr = reinterpret_cast<unsigned int *>(zero + PEP_ZERO_TILES) + (size_t)s * 8 * 32;          // (cleared by the search's one fill)
            if (stride_lookup) {
                const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(T.total, NTILE), 256u * 8u);
                a.tile_groups = grid >= 64 ? 8u : 1u;
                hipLaunchKernelGGL(seed_match_stride, dim3(grid), dim3(256), 0, ctx->stream, a);
            } else {
                const unsigned grid = std::min(tb, 256u * 8u);          // (with claimed tiles 8 .. 16 blocks per compute unit take the same time: profiles/r06_block_times.txt)
                a.tile_groups = grid >= 64 ? 8u : 1u;
                PEP_SEED_DISPATCH(seed_match, dim3(grid), sh, a);
            }
            pep_timer_end(ctx, TM_MATCH0 + s);
#ifdef PEP_PROBES
            if (P.reserved[0] == 7 && use_partition && sh.weight == 10) {
                // measurement only: the scatter pass of a partitioned join over the same targets, behind the matcher it would replace
                static DevBuf probe_part, probe_cnt;
                const uint32_t n_coarse = 1u << (bucket_bits - fine_bits), cap = 16384;
                PEP_TRY(dev_reserve(ctx, probe_part, (uint64_t)n_coarse * cap * 8));
                PEP_TRY(dev_reserve(ctx, probe_cnt, (uint64_t)n_coarse * 4 + 64));
                PEP_HIP(ctx, hipMemsetAsync(probe_cnt.p, 0, (uint64_t)n_coarse * 4 + 64, ctx->stream));
                const int tiles = (int)std::max<uint64_t>(PART_TILES, ceil_div(T.total, (uint64_t)TILE * 8192));
                const unsigned pb = (unsigned)ceil_div(T.total, (uint64_t)tiles * TILE);
                hipLaunchKernelGGL(tgt_slab_probe<10>, dim3(pb), dim3(256), (size_t)n_coarse * 4 + TILE + TILE_HALO, ctx->stream, sh, T.res.as<const uint8_t>(), T.total, bucket_bits, fine_bits,
                                   (const unsigned long long *)filter, probe_cnt.as<uint32_t>(), probe_part.as<uint64_t>(), cap, tiles,
                                   reinterpret_cast<unsigned long long *>(probe_cnt.as<uint32_t>() + n_coarse));
            }
#endif
            // (same-box A/B, tools/ab/phase2_ab.sh: four trips of 64 hits per strip; 8 blocks per CU at 10 k genes - 122 us against 126 with 16 -, 16 at 50 k - 1.71 ms against 1.86)
            hipLaunchKernelGGL(seed_runs_extend, dim3(256u * (T.total > (48ull << 20) ? 16u : 8u)), dim3(256), 0, ctx->stream, a);
            PEP_HIP(ctx, hipGetLastError());
        }
        // field widths of the dense key form (see keys_pack)
        const uint32_t bin_min = (uint32_t)(((1 << 23) - (int)std::max<uint32_t>(Q.max_len, 1u) + 1) >> 6);
        const uint32_t bin_max = (uint32_t)(((1 << 23) + (int)std::max<uint32_t>(T.max_len, 1u) - 1) >> 6);
        const int tb = std::max(1, ilog2_ceil(T.n)), qb = std::max(1, ilog2_ceil(Q.n)), bb = std::max(1, ilog2_ceil((uint64_t)bin_max - bin_min + 1));
        // (the histogram of the keys' top digit rides along: the two-level sort below needs it, and the host decides from it)
        const int key_bits = qb + tb + bb, top_shift = key_bits > PEP_SORT_TOP_BITS && key_bits <= 54 && P.reserved[2] == 0 ? key_bits - PEP_SORT_TOP_BITS : -1;
        uint32_t *top_hist = reinterpret_cast<uint32_t *>(zero + PEP_ZERO_TOP);
        hipLaunchKernelGGL(set_compact, dim3((unsigned)ceil_div(cap, 256 * COMPACT_ROUNDS)), dim3(256), 0, ctx->stream, ctx->d_set.as<uint64_t>(), cap,
                           ctx->ws[4].as<uint64_t>(), list_cap, counters, tb, bb, bin_min, top_hist, top_shift);
        ctx->set_clean_slots = cap;
        struct { uint32_t counters[4]; unsigned long long stats[3]; uint32_t n_entries[4]; uint32_t pad[2]; uint32_t top[1 << PEP_SORT_TOP_BITS]; } h_all;
        // counters[0..3], the three statistics words, the index sizes per shape and the top-digit histogram of the candidate keys: one copy
        static_assert(sizeof(h_all) == 64 + 4096 && PEP_ZERO_TOP == PEP_ZERO_SEED + 64, "layout of the counter block");
        if (before_sync) { PEP_TRY(before_sync(ctx)); before_sync = nullptr; }      // (once, also when the stage is repeated with larger buffers)
        // the counters go to the host and whatever before_sync wants on the device (the score thresholds) comes up, in ONE kernel over pinned memory
        PEP_TRY(pep_read_back_with_upload(ctx, &h_all, counters, sizeof(h_all), ctx->upload.d_dst, ctx->upload.pinned_src, ctx->upload.n_words));
        ctx->upload.n_words = 0;
        PEP_TRY(pep_sync_reads(ctx));
        const uint32_t *h_counters = h_all.counters;
        const unsigned long long *h_stats = h_all.stats;
        if (h_counters[3]) { use_partition = false; continue; }       // a coarse index bucket did not fit LDS: rebuild the plain way
        if (h_counters[2]) { hit_cap *= 4; continue; }                // raw hit buffer too small: retry 4x larger
        if (h_counters[1] || h_counters[0] > list_cap) { table_bits += 2; continue; }     // set too small: retry 4x larger
        for (int s = 0; s < P.n_shapes; ++s) q_seeds += h_all.n_entries[s];
        ctx->stats.query_seeds = q_seeds;
        ctx->stats.target_seeds = h_stats[0];
        ctx->stats.seed_hits = h_stats[1];
        ctx->stats.seed_hits_passed = h_stats[2];
        const uint64_t n = h_counters[0];
        if (n) {
            pep_key_unpack up;
            up.on = 1; up.tb = tb; up.bb = bb; up.bin_min = bin_min;
            void *sort_hist = nullptr;
            uint32_t top_max = 0;
            for (uint32_t v : h_all.top) top_max = std::max(top_max, v);
            if (top_shift >= 0 && top_max <= PEP_SORT_TOP_CAP) {
                // every bucket of the top digit fits LDS: partition + per-bucket sort, two launches
                PEP_TRY(pep_sort_u64_two_level(ctx, ctx->ws[4].as<uint64_t>(), ctx->ws[5].as<uint64_t>(), n, key_bits, top_hist, &up));
                *d_cands = ctx->ws[4].as<uint64_t>();
                *n_cands = n;
                return PEP_OK;
            }
            PEP_TRY(pep_zero_block(ctx, PEP_ZC_SORT, PEP_ZERO_SORT, 8 * 2048 * 4, &sort_hist));
            PEP_TRY(pep_sort_u64(ctx, ctx->ws[4].as<uint64_t>(), ctx->ws[5].as<uint64_t>(), nullptr, n, qb + tb + bb, reinterpret_cast<uint32_t *>(sort_hist), &up));
        }
        *d_cands = ctx->ws[4].as<uint64_t>();
        *n_cands = n;
        return PEP_OK;
    }
    return pep_fail(ctx, PEP_ERR_LIMIT, "candidate hash set overflow after 8 growth attempts");
}
