// LSD radix sort of u64 keys (stable), used for the candidate list (K4).
//
// ONE launch per pass ("onesweep"): a tile of 8192 keys counts its digits in LDS, publishes the counts, finds the number of equal digits
// in the tiles before it by decoupled look-back (one status word per (tile, digit); a thread follows the chain of its own digits),
// and scatters its keys behind them in a stable order.  The digit bases (keys with a smaller digit, anywhere) come from global
// histograms that ONE launch in front computes for all passes at once.  The sort of ~10^5 candidate keys is bound by launches, not by
// bytes: hist -> scan -> scatter per pass was 12 launches for the four passes of a search, this is 5.
// The digit width is chosen per call: the fewest passes (an even number, so that the result lands in the input buffer without a
// copy) with digits of at most 11 bits - the candidate keys of a search are 35-40 bits wide: four passes of 9-10 bits.
// Optionally the last pass rewrites the keys on their way out (KeyUnpack: the dense candidate form -> q:21 | t:25 | bin:18).
#include "common.h"

namespace {

constexpr int ST = 1024;                // threads per block: a tile is 8192 keys, so that the look-back chain of a search's ~50 k candidate keys is six tiles long
constexpr int SI = 8;                   // keys per thread
constexpr int STILE = ST * SI;
constexpr int MAX_DIGIT_BITS = 11;
constexpr int MAX_PASSES = 8;

// status word: epoch << 34 | flag << 32 | value (flag 1 = the tile's own count, 2 = count of this digit up to and including the tile).
// The epoch is a per-pass number: words left by earlier passes never match, so nothing is cleared between them.
constexpr uint64_t F_SUM = 1, F_PREFIX = 2;

struct HistArgs { int shift[MAX_PASSES]; int passes; };

// global digit histograms of all passes in one read of the keys: hist[p][d]  (zeroed by the caller)
template <int DB>
__global__ __launch_bounds__(ST) void sort_hist_all(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ n_ptr, uint64_t n_host, uint32_t *__restrict__ hist, HistArgs a)
{
    constexpr uint32_t R = 1u << DB;
    extern __shared__ uint32_t h[];                     // [passes][R]
    const uint64_t n = n_ptr ? (uint64_t)*n_ptr : n_host;
    for (uint32_t d = threadIdx.x; d < R * (uint32_t)a.passes; d += ST) h[d] = 0;
    __syncthreads();
    for (uint64_t base = (uint64_t)blockIdx.x * STILE; base < n; base += (uint64_t)gridDim.x * STILE) {
#pragma unroll
        for (int r = 0; r < SI; ++r) {
            const uint64_t i = base + (uint64_t)r * ST + threadIdx.x;
            if (i < n) {
                const uint64_t k = keys[i];
                for (int p = 0; p < a.passes; ++p) atomicAdd(&h[p * R + ((k >> a.shift[p]) & (R - 1))], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < R * (uint32_t)a.passes; d += ST) if (h[d]) atomicAdd(&hist[d], h[d]);
}

// ---- two-level sort for UNIQUE keys (the candidate list of a search): one onesweep pass on the TOP 10 bits of the key, then every bucket of
// that digit is sorted inside LDS by one block (bitonic) - two launches instead of five for the ~50 k keys of a search, whose sort is
// bound by launches and barrier chains, not by bytes.  The histogram of the top digit comes from the kernel that produced the keys
// (set_compact) and travels to the host with the search's counters: the host takes this path only when every bucket fits (TOP_CAP keys).
__global__ __launch_bounds__(256) void sort_buckets(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, const uint32_t *__restrict__ top_hist, pep_key_unpack unpack)
{
    __shared__ uint64_t k[PEP_SORT_TOP_CAP];
    __shared__ uint32_t s_part[4];
    const uint32_t b = blockIdx.x, cnt = top_hist[b];
    if (cnt == 0) return;                                    // (block-uniform)
    // keys of the buckets in front of this one
    uint32_t before = 0;
    for (uint32_t d = threadIdx.x; d < b; d += 256) before += top_hist[d];
    for (int x = 32; x > 0; x >>= 1) before += __shfl_xor(before, x, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = before;
    __syncthreads();
    const uint32_t start = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    uint32_t m = 1;
    while (m < cnt) m <<= 1;
    for (uint32_t x = threadIdx.x; x < m; x += 256) k[x] = x < cnt ? in[start + x] : ~0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= m; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < (m >> 1); t += 256) {
                const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;           // the pair (lo, hi) of this step
                const bool up = (lo & size) == 0;
                const uint64_t a = k[lo], c = k[hi];
                if ((a > c) == up) { k[lo] = c; k[hi] = a; }
            }
            __syncthreads();
        }
    for (uint32_t x = threadIdx.x; x < cnt; x += 256) {
        uint64_t v = k[x];
        if (unpack.on) {
            const uint64_t q = v >> (unpack.tb + unpack.bb), t = (v >> unpack.bb) & ((1ull << unpack.tb) - 1), bin = (v & ((1ull << unpack.bb) - 1)) + unpack.bin_min;
            v = (q << 43) | (t << 18) | bin;
        }
        out[start + x] = v;
    }
}

template <int DB>
__global__ __launch_bounds__(ST) void sort_onesweep(const uint64_t *__restrict__ keys, uint64_t *__restrict__ out, const uint32_t *__restrict__ n_ptr, uint64_t n_host,
                                                    int shift, const uint32_t *__restrict__ hist /* [R] of this pass */, uint64_t *__restrict__ state,
                                                    uint32_t ticket_base, uint64_t epoch, pep_key_unpack unpack)
{
    constexpr uint32_t R = 1u << DB;
    __shared__ __attribute__((aligned(16))) uint8_t wave_cnt[ST / 64][R];       // keys of one digit in one wavefront's row: at most 64
    __shared__ uint32_t digit_base[R];
    __shared__ uint32_t local[R];
    __shared__ uint32_t s_tile, s_scan[ST / 64];
    const uint64_t n = n_ptr ? (uint64_t)*n_ptr : n_host;
    const uint32_t nb = (uint32_t)((n + STILE - 1) / STILE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = atomicAdd(reinterpret_cast<uint32_t *>(state), 1u) - ticket_base;
    for (uint32_t d = threadIdx.x; d < R; d += ST) local[d] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= nb) return;                              // surplus blocks of a grid sized from an upper bound (block-uniform)
    uint64_t *status = state + 1;
    const uint64_t base = (uint64_t)tile * STILE;
    uint64_t key[SI];
#pragma unroll
    for (int r = 0; r < SI; ++r) {
        const uint64_t i = base + (uint64_t)r * ST + threadIdx.x;
        key[r] = i < n ? keys[i] : ~0ull;
        if (i < n) atomicAdd(&local[(uint32_t)(key[r] >> shift) & (R - 1)], 1u);
    }
    __syncthreads();
    // exclusive scan of the global histogram over the digits (digit d = thread d; R <= 2048 = two per thread at most): keys with a smaller
    // digit, anywhere
    {
        constexpr int PER = R > (uint32_t)ST ? (int)(R / ST) : 1;
        uint32_t own[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { const uint32_t d = threadIdx.x * PER + k; own[k] = d < R ? hist[d] : 0u; sum += own[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w = 0; w < wave; ++w) run += s_scan[w];
#pragma unroll
        for (int k = 0; k < PER; ++k) { const uint32_t d = threadIdx.x * PER + k; if (d < R) digit_base[d] = run; run += own[k]; }
        __syncthreads();
    }
    // look-back per digit: thread t follows digits t, t + 256, ... (coalesced status rows)
    for (uint32_t d = threadIdx.x; d < R; d += ST) {
        const uint32_t mine = local[d];
        uint32_t before = 0;
        if (tile > 0) {
            __hip_atomic_store(&status[(uint64_t)tile * R + d], (epoch << 34) | (F_SUM << 32) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t t = (int64_t)tile - 1; t >= 0; --t) {
                uint64_t w;
                do { w = __hip_atomic_load(&status[(uint64_t)t * R + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> 34) != epoch);
                before += (uint32_t)w;
                if (((w >> 32) & 3u) == F_PREFIX) break;
            }
        }
        __hip_atomic_store(&status[(uint64_t)tile * R + d], (epoch << 34) | (F_PREFIX << 32) | (before + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        digit_base[d] += before;
    }
    __syncthreads();
    // stable scatter, one row of 1024 keys at a time: rank inside the wavefront by ballots, wavefronts in order
    for (int r = 0; r < SI; ++r) {
        for (uint32_t x = threadIdx.x; x < (ST / 64) * R / 4; x += ST) reinterpret_cast<uint32_t *>(&wave_cnt[0][0])[x] = 0u;
        __syncthreads();
        const uint64_t i = base + (uint64_t)r * ST + threadIdx.x;
        const bool valid = i < n;
        const uint64_t k = key[r];
        const uint32_t d = (uint32_t)(k >> shift) & (R - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        if (valid && rank == 0) wave_cnt[wave][d] = (uint8_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t o = digit_base[d] + rank;
            for (int w = 0; w < wave; ++w) o += wave_cnt[w][d];
            uint64_t v = k;
            if (unpack.on) {
                const uint64_t q = v >> (unpack.tb + unpack.bb), t = (v >> unpack.bb) & ((1ull << unpack.tb) - 1), bin = (v & ((1ull << unpack.bb) - 1)) + unpack.bin_min;
                v = (q << 43) | (t << 18) | bin;
            }
            out[o] = v;
        }
        __syncthreads();
        for (uint32_t dd = threadIdx.x; dd < R; dd += ST) {
            uint32_t add = 0;
#pragma unroll
            for (int w = 0; w < ST / 64; ++w) add += wave_cnt[w][dd];
            digit_base[dd] += add;
        }
        __syncthreads();
    }
}

template <int DB>
int sort_passes(pep_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, const uint32_t *d_n, uint64_t n_bound, int bits, int passes, uint32_t *d_hist_zeroed,
                const pep_key_unpack *unpack)
{
    constexpr uint32_t R = 1u << DB;
    const uint32_t nb = (uint32_t)ceil_div(n_bound, STILE);
    DevBuf &S = ctx->sort_state;
    const size_t need = ((size_t)nb * R + 2) * sizeof(uint64_t);
    bool clear = false;
    if (need > S.cap) { PEP_TRY(dev_reserve(ctx, S, need)); clear = true; }
    DevBuf &H = ctx->sort_hist;
    uint32_t *hist = d_hist_zeroed;
    if (!hist) {
        PEP_TRY(dev_reserve(ctx, H, (size_t)MAX_PASSES * (1u << MAX_DIGIT_BITS) * 4));
        PEP_HIP(ctx, hipMemsetAsync(H.p, 0, (size_t)passes * R * 4, ctx->stream));
        hist = H.as<uint32_t>();
    }
    HistArgs ha;
    ha.passes = 0;
    for (int p = 0; p < passes; ++p) ha.shift[ha.passes++] = p * DB;        // (a pass beyond the key width sees digit 0 everywhere: a stable copy)
    hipLaunchKernelGGL(sort_hist_all<DB>, dim3(std::min<uint32_t>(nb, 1024u)), dim3(ST), (size_t)ha.passes * R * 4, ctx->stream, (const uint64_t *)d_keys, d_n, n_bound, hist, ha);
    uint64_t *src = d_keys, *dst = d_tmp;
    int real = 0;
    pep_key_unpack none;
    none.on = 0; none.tb = none.bb = 0; none.bin_min = 0;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * DB;
        const bool last = p == passes - 1;
        {
            if (ctx->sort_dirty) { clear = true; ctx->sort_dirty = false; }
            ctx->sort_epoch = (ctx->sort_epoch + 1) & ((1u << 29) - 1);
            if (ctx->sort_epoch == 0) { clear = true; ctx->sort_epoch = 1; }          // wrapped: forget every old word
            if (clear) { PEP_HIP(ctx, hipMemsetAsync(S.p, 0, S.cap, ctx->stream)); ctx->sort_ticket_base = 0; clear = false; }
            hipLaunchKernelGGL(sort_onesweep<DB>, dim3(nb), dim3(ST), 0, ctx->stream, (const uint64_t *)src, dst, d_n, n_bound, shift, (const uint32_t *)(hist + (size_t)real * R),
                               S.as<uint64_t>(), ctx->sort_ticket_base, (uint64_t)ctx->sort_epoch, (last && unpack) ? *unpack : none);
            ctx->sort_ticket_base += nb;
            ++real;
        }
        uint64_t *t = src; src = dst; dst = t;
    }
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

}  // namespace

// Sorts d_keys ascending on their low `bits` bits; d_tmp has room for the keys; the result is in d_keys.  The number of keys is *d_n
// when d_n is given (device memory; n_bound then only sizes the grids and must not be smaller), n_bound otherwise.
// d_hist_zeroed: MAX_PASSES x 2^11 zeroed u32 words the caller provides (nullptr: the sort clears its own).
int pep_sort_u64(pep_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, const uint32_t *d_n, uint64_t n_bound, int bits, uint32_t *d_hist_zeroed, const pep_key_unpack *unpack)
{
    if (n_bound < 2 && !d_n) return PEP_OK;
    if (n_bound == 0) return PEP_OK;
    if (n_bound >= (1ull << 30)) return pep_fail(ctx, PEP_ERR_LIMIT, "pep_sort_u64: more than 2^30 keys");
    bits = std::max(1, std::min(64, bits));
    int passes = 2;
    while ((bits + passes - 1) / passes > MAX_DIGIT_BITS) passes += 2;
    if (passes > MAX_PASSES) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_sort_u64: too many passes");
    const int db = std::max(8, (bits + passes - 1) / passes);
    if ((passes - 1) * db >= 64) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_sort_u64: pass beyond 64 bits");
    switch (db) {
        case 8: return sort_passes<8>(ctx, d_keys, d_tmp, d_n, n_bound, bits, passes, d_hist_zeroed, unpack);
        case 9: return sort_passes<9>(ctx, d_keys, d_tmp, d_n, n_bound, bits, passes, d_hist_zeroed, unpack);
        case 10: return sort_passes<10>(ctx, d_keys, d_tmp, d_n, n_bound, bits, passes, d_hist_zeroed, unpack);
        default: return sort_passes<11>(ctx, d_keys, d_tmp, d_n, n_bound, bits, passes, d_hist_zeroed, unpack);
    }
}

// the two-level sort (see sort_buckets): n distinct keys of `bits` significant bits, d_top_hist = the histogram of their top
// PEP_SORT_TOP_BITS bits (device memory), every count of which the caller has seen to be <= PEP_SORT_TOP_CAP.  Result in d_keys.
int pep_sort_u64_two_level(pep_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint64_t n, int bits, const uint32_t *d_top_hist, const pep_key_unpack *unpack)
{
    if (n == 0) return PEP_OK;
    if (bits <= PEP_SORT_TOP_BITS || n >= (1ull << 30)) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_sort_u64_two_level: key width / count");
    constexpr int DB = PEP_SORT_TOP_BITS;
    constexpr uint32_t R = 1u << DB;
    const uint32_t nb = (uint32_t)ceil_div(n, STILE);
    DevBuf &S = ctx->sort_state;
    const size_t need = ((size_t)nb * R + 2) * sizeof(uint64_t);
    bool clear = false;
    if (need > S.cap) { PEP_TRY(dev_reserve(ctx, S, need)); clear = true; }
    if (ctx->sort_dirty) { clear = true; ctx->sort_dirty = false; }
    ctx->sort_epoch = (ctx->sort_epoch + 1) & ((1u << 29) - 1);
    if (ctx->sort_epoch == 0) { clear = true; ctx->sort_epoch = 1; }
    if (clear) { PEP_HIP(ctx, hipMemsetAsync(S.p, 0, S.cap, ctx->stream)); ctx->sort_ticket_base = 0; }
    pep_key_unpack none;
    none.on = 0; none.tb = none.bb = 0; none.bin_min = 0;
    hipLaunchKernelGGL(sort_onesweep<DB>, dim3(nb), dim3(ST), 0, ctx->stream, (const uint64_t *)d_keys, d_tmp, (const uint32_t *)nullptr, n, bits - DB, d_top_hist,
                       S.as<uint64_t>(), ctx->sort_ticket_base, (uint64_t)ctx->sort_epoch, none);
    ctx->sort_ticket_base += nb;
    hipLaunchKernelGGL(sort_buckets, dim3(R), dim3(256), 0, ctx->stream, (const uint64_t *)d_tmp, d_keys, d_top_hist, unpack ? *unpack : none);
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}
