// K2-K4: spaced seeds over a reduced alphabet, bucketed query index, streaming join of the target
// seeds against it, de-duplication of (query, target, diagonal bin) candidates.
//
//  seed_count / seed_fill : one thread per packed query byte position; key = sum red[r[p+off_k]] * base^k.
//                           Padding bytes (code 31) never seed, so a seed cannot straddle two sequences.
//                           Index = counting sort by hash(key): counts -> exclusive scan -> fill (8 B entries
//                           key << 29 | pos).  Algorithmic traffic 1 B read + 8 B written per query residue.
//  seed_join              : one thread per packed target byte position; reads its bucket (start/end = 8 B,
//                           entries 8 B each), and for every equal key inserts the candidate key
//                           q:21 | t:25 | bin:18 into a device hash set; first inserter appends it to the list.
//  The candidate list is then radix-sorted (sort.hip) so every later stage is order-deterministic.
#include "common.h"

namespace {

struct SeedShape {
    int32_t weight;
    int32_t base;
    int32_t offs[32];
    uint8_t reduce[32];
};

constexpr uint64_t EMPTY = ~0ull;
constexpr int POS_BITS = 29;
constexpr uint64_t POS_MASK = (1ull << POS_BITS) - 1;

__device__ __forceinline__ uint32_t hash_u64(uint64_t k, int bits)
{
    return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> (64 - bits));
}

__device__ __forceinline__ bool seed_key(const SeedShape &sh, const uint8_t *__restrict__ res, uint64_t p, uint64_t &key)
{
    uint64_t k = 0, mul = 1;
    bool ok = true;
#pragma unroll 1
    for (int i = 0; i < sh.weight; ++i) {
        const uint8_t c = res[p + sh.offs[i]];
        const uint8_t g = sh.reduce[c & 31];
        ok = ok && (g != 0xFF);
        k += mul * g;
        mul *= (uint64_t)sh.base;
    }
    key = k;
    return ok;
}

// largest i in [0, n) with off[i] <= p   (off[n] is a sentinel > every position)
__device__ __forceinline__ uint32_t find_seq(const uint32_t *__restrict__ off, uint32_t n, uint32_t p)
{
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void seed_count(SeedShape sh, const uint8_t *__restrict__ res, uint64_t total, uint32_t *__restrict__ cnt, int bucket_bits)
{
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p + 32 > total) return;             // the trailing PEP_END_PAD bytes hold no residues
    uint64_t key;
    if (seed_key(sh, res, p, key)) atomicAdd(&cnt[hash_u64(key, bucket_bits)], 1u);
}

__global__ __launch_bounds__(256) void seed_fill(SeedShape sh, const uint8_t *__restrict__ res, uint64_t total, const uint32_t *__restrict__ start,
                                                 uint32_t *__restrict__ fill, uint64_t *__restrict__ entries, int bucket_bits)
{
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p + 32 > total) return;
    uint64_t key;
    if (seed_key(sh, res, p, key)) {
        const uint32_t b = hash_u64(key, bucket_bits);
        const uint32_t slot = start[b] + atomicAdd(&fill[b], 1u);
        entries[slot] = (key << POS_BITS) | p;
    }
}

struct JoinArgs {
    const uint8_t *t_res;
    uint64_t t_total;
    const uint32_t *t_off;
    uint32_t nt;
    const uint32_t *q_off;
    uint32_t nq;
    const uint32_t *start;
    const uint64_t *entries;
    int bucket_bits;
    uint64_t *table;
    int table_bits;
    uint64_t *list;
    uint32_t list_cap;
    uint32_t *counters;      // [0] = list length, [1] = overflow flag
    unsigned long long *stats;   // [0] = target seeds, [1] = seed hits
};

__device__ __forceinline__ void set_insert(const JoinArgs &a, uint64_t k)
{
    const uint32_t mask = (1u << a.table_bits) - 1;
    uint32_t slot = hash_u64(k, a.table_bits);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const uint64_t cur = a.table[slot];
        if (cur == k) return;
        if (cur == EMPTY) {
            const uint64_t old = atomicCAS((unsigned long long *)&a.table[slot], (unsigned long long)EMPTY, (unsigned long long)k);
            if (old == k) return;
            if (old == EMPTY) {
                const uint32_t idx = atomicAdd(&a.counters[0], 1u);
                if (idx < a.list_cap) a.list[idx] = k; else a.counters[1] = 1u;
                return;
            }
        }
        slot = (slot + 1) & mask;
    }
    a.counters[1] = 1u;
}

__global__ __launch_bounds__(256) void seed_join(SeedShape sh, JoinArgs a)
{
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_seed = 0, n_hit = 0;
    uint64_t key;
    if (p + 32 <= a.t_total && seed_key(sh, a.t_res, p, key)) {
        n_seed = 1;
        const uint32_t b = hash_u64(key, a.bucket_bits);
        const uint32_t e0 = a.start[b], e1 = a.start[b + 1];
        uint32_t t = 0xFFFFFFFFu, tpos = 0;
        for (uint32_t e = e0; e < e1; ++e) {
            const uint64_t ent = a.entries[e];
            if ((ent >> POS_BITS) != key) continue;
            ++n_hit;
            if (t == 0xFFFFFFFFu) {
                t = find_seq(a.t_off, a.nt, (uint32_t)p);
                tpos = (uint32_t)p - a.t_off[t];
            }
            const uint32_t qp = (uint32_t)(ent & POS_MASK);
            const uint32_t q = find_seq(a.q_off, a.nq, qp);
            const int32_t diag = (int32_t)tpos - (int32_t)(qp - a.q_off[q]);
            const uint32_t bin = (uint32_t)(diag + (1 << 23)) >> 6;
            set_insert(a, ((uint64_t)q << 43) | ((uint64_t)t << 18) | (uint64_t)bin);
        }
    }
    // per-wave statistics: one atomic per wave
    for (int d = 32; d > 0; d >>= 1) {
        n_seed += __shfl_down(n_seed, d, 64);
        n_hit += __shfl_down(n_hit, d, 64);
    }
    if ((threadIdx.x & 63) == 0 && (n_seed | n_hit)) {
        atomicAdd(&a.stats[0], (unsigned long long)n_seed);
        atomicAdd(&a.stats[1], (unsigned long long)n_hit);
    }
}

int ilog2_ceil(uint64_t x)
{
    int b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

}  // namespace

// workspace slots used here: ws[0] cnt, ws[1] start, ws[2] entries, ws[3] table, ws[4] list, ws[5] list tmp (sort),
// ws[6] counters+stats, ws[7] scan scratch, ws[8] sort histogram
int pep_find_candidates(pep_ctx *ctx, uint64_t **d_cands, uint64_t *n_cands)
{
    const pep_search_params &P = ctx->params;
    SeqSet &Q = ctx->q, &T = ctx->t;
    if (Q.total > PEP_MAX_RESIDUES || T.total > PEP_MAX_RESIDUES) return pep_fail(ctx, PEP_ERR_LIMIT, "more than 2^29 packed residues on one side");
    *n_cands = 0;
    *d_cands = nullptr;
    ctx->stats.query_seeds = ctx->stats.target_seeds = ctx->stats.seed_hits = 0;
    if (Q.n == 0 || T.n == 0) return PEP_OK;

    const int bucket_bits = std::max(10, std::min(28, ilog2_ceil(2 * Q.total)));
    const uint64_t n_buckets = 1ull << bucket_bits;
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], (n_buckets + 1) * sizeof(uint32_t)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[1], (n_buckets + 2) * sizeof(uint32_t)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[2], (Q.total + 1) * sizeof(uint64_t)));
    PEP_TRY(dev_reserve(ctx, ctx->ws[6], 64));
    uint32_t *cnt = ctx->ws[0].as<uint32_t>(), *start = ctx->ws[1].as<uint32_t>();
    uint64_t *entries = ctx->ws[2].as<uint64_t>();
    uint32_t *counters = ctx->ws[6].as<uint32_t>();
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(counters + 4);

    int table_bits = 22;
    for (int attempt = 0; attempt < 8; ++attempt, table_bits += 2) {
        const uint64_t cap = 1ull << table_bits;
        const uint32_t list_cap = (uint32_t)(cap >> 1);
        PEP_TRY(dev_reserve(ctx, ctx->ws[3], cap * sizeof(uint64_t)));
        PEP_TRY(dev_reserve(ctx, ctx->ws[4], (uint64_t)list_cap * sizeof(uint64_t)));
        PEP_TRY(dev_reserve(ctx, ctx->ws[5], (uint64_t)list_cap * sizeof(uint64_t)));
        PEP_HIP(ctx, hipMemsetAsync(ctx->ws[3].p, 0xFF, cap * sizeof(uint64_t), ctx->stream));
        PEP_HIP(ctx, hipMemsetAsync(counters, 0, 64, ctx->stream));
        uint64_t q_seeds = 0;
        for (int s = 0; s < P.n_shapes; ++s) {
            SeedShape sh;
            sh.weight = P.weight[s];
            sh.base = P.base;
            for (int i = 0; i < 32; ++i) { sh.offs[i] = P.offs[s][i]; sh.reduce[i] = P.reduce[i]; }
            PEP_HIP(ctx, hipMemsetAsync(cnt, 0, (n_buckets + 1) * sizeof(uint32_t), ctx->stream));
            const unsigned qb = (unsigned)ceil_div(Q.total, 256), tb = (unsigned)ceil_div(T.total, 256);
            hipLaunchKernelGGL(seed_count, dim3(qb), dim3(256), 0, ctx->stream, sh, Q.res.as<const uint8_t>(), Q.total, cnt, bucket_bits);
            PEP_TRY(pep_scan_u32(ctx, cnt, start, n_buckets, ctx->ws[7]));
            PEP_HIP(ctx, hipMemsetAsync(cnt, 0, (n_buckets + 1) * sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(seed_fill, dim3(qb), dim3(256), 0, ctx->stream, sh, Q.res.as<const uint8_t>(), Q.total, (const uint32_t *)start, cnt, entries, bucket_bits);
            JoinArgs a;
            a.t_res = T.res.as<const uint8_t>(); a.t_total = T.total; a.t_off = T.off.as<const uint32_t>(); a.nt = T.n;
            a.q_off = Q.off.as<const uint32_t>(); a.nq = Q.n; a.start = start; a.entries = entries; a.bucket_bits = bucket_bits;
            a.table = ctx->ws[3].as<uint64_t>(); a.table_bits = table_bits; a.list = ctx->ws[4].as<uint64_t>(); a.list_cap = list_cap;
            a.counters = counters; a.stats = stats;
            hipLaunchKernelGGL(seed_join, dim3(tb), dim3(256), 0, ctx->stream, sh, a);
            PEP_HIP(ctx, hipGetLastError());
            uint32_t nseed = 0;
            PEP_HIP(ctx, hipMemcpyAsync(&nseed, start + n_buckets, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            q_seeds += nseed;
        }
        uint32_t h_counters[4];
        unsigned long long h_stats[2];
        PEP_HIP(ctx, hipMemcpyAsync(h_counters, counters, sizeof(h_counters), hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, hipMemcpyAsync(h_stats, stats, sizeof(h_stats), hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h_counters[1] || h_counters[0] > list_cap) continue;     // table too small: retry 4x larger
        ctx->stats.query_seeds = q_seeds;
        ctx->stats.target_seeds = h_stats[0];
        ctx->stats.seed_hits = h_stats[1];
        const uint64_t n = h_counters[0];
        PEP_TRY(pep_sort_u64(ctx, ctx->ws[4].as<uint64_t>(), ctx->ws[5].as<uint64_t>(), n, 64, ctx->ws[8]));
        *d_cands = ctx->ws[4].as<uint64_t>();
        *n_cands = n;
        return PEP_OK;
    }
    return pep_fail(ctx, PEP_ERR_LIMIT, "candidate hash set overflow after 8 growth attempts");
}
