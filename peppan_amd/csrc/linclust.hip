// K9: linear-time clustering that replaces `mmseqs createdb / linclust / createtsv` (clust.py:62-66).
// Algorithm = oracle_linclust (oracle/align_oracle.c): min-hash k-mer selection, centre = longest sequence
// per k-mer, ungapped verification of each member on its k-mer diagonal, greedy assignment by priority.
//
//   lc_select  one wavefront per sequence: m rounds of "smallest (hash, pos) above the previous pick"      (HBM: L bytes read m times from L1/L2)
//   lc_insert  one thread per selected k-mer: open-addressing map key -> atomicMax(len << 32 | ~idx)       (random 16 B)
//   lc_verify  one wavefront per sequence: centre look-up, diagonal, lanes stride over the overlap          (HBM: 2 x overlap bytes per pair)
//   lc_assign  rounds of a monotone fixed point that equals the sequential greedy assignment                (random 4 B)
// All integer except the two threshold comparisons, done in IEEE double exactly as the oracle does.
#include "common.h"

namespace {

constexpr uint64_t EMPTY_KEY = ~0ull;
constexpr int MAX_M = 32;

__device__ __forceinline__ uint64_t lc_mix(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ bool kmer_at(const uint8_t *__restrict__ s, uint32_t p, int base, int k, uint64_t &key)
{
    uint64_t v = 0, mul = 1;
    bool ok = true;
    for (int i = 0; i < k; ++i) {
        const uint8_t c = s[p + i];
        ok = ok && (c < base);
        v += mul * c;
        mul *= (uint64_t)base;
    }
    key = v;
    return ok;
}

// (h, pos) lexicographic "less"
__device__ __forceinline__ bool hp_less(uint64_t h1, uint32_t p1, uint64_t h2, uint32_t p2) { return h1 < h2 || (h1 == h2 && p1 < p2); }

__global__ __launch_bounds__(256) void lc_select(const uint8_t *__restrict__ res, const uint64_t *__restrict__ off, uint32_t n, int base, int k, int m,
                                                 uint64_t *__restrict__ sel_key, uint32_t *__restrict__ sel_pos, uint32_t *__restrict__ sel_cnt)
{
    const int lane = threadIdx.x & 63;
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n) return;
    const uint8_t *q = res + off[s];
    const uint32_t L = (uint32_t)(off[s + 1] - off[s]);
    uint64_t last_h = 0;
    uint32_t last_p = 0, cnt = 0;
    bool first = true;
    for (int r = 0; r < m; ++r) {
        uint64_t bh = ~0ull, bkey = 0;
        uint32_t bp = 0xFFFFFFFFu;
        for (uint32_t p = lane; p + k <= L; p += 64) {
            uint64_t key;
            if (!kmer_at(q, p, base, k, key)) continue;
            const uint64_t h = lc_mix(key);
            if (!first && !hp_less(last_h, last_p, h, p)) continue;          // already taken
            if (hp_less(h, p, bh, bp)) { bh = h; bp = p; bkey = key; }
        }
        for (int d = 32; d > 0; d >>= 1) {
            const uint64_t oh = __shfl_xor(bh, d, 64), ok = __shfl_xor(bkey, d, 64);
            const uint32_t op = __shfl_xor(bp, d, 64);
            if (hp_less(oh, op, bh, bp)) { bh = oh; bp = op; bkey = ok; }
        }
        if (bp == 0xFFFFFFFFu) break;
        if (lane == 0) { sel_key[(uint64_t)s * m + r] = bkey; sel_pos[(uint64_t)s * m + r] = bp; }
        last_h = bh; last_p = bp; first = false;
        ++cnt;
    }
    if (lane == 0) sel_cnt[s] = cnt;
}

__device__ __forceinline__ uint32_t map_slot(uint64_t key, int bits) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - bits)); }

__global__ __launch_bounds__(256) void lc_insert(const uint64_t *__restrict__ off, uint32_t n, int m, const uint64_t *__restrict__ sel_key,
                                                 const uint32_t *__restrict__ sel_cnt, uint64_t *__restrict__ map_key, unsigned long long *__restrict__ map_val,
                                                 int bits, uint32_t *__restrict__ overflow)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (uint64_t)n * m) return;
    const uint32_t s = (uint32_t)(e / m), r = (uint32_t)(e % m);
    if (r >= sel_cnt[s]) return;
    const uint64_t key = sel_key[e];
    const unsigned long long val = ((unsigned long long)(uint32_t)(off[s + 1] - off[s]) << 32) | (unsigned long long)(0xFFFFFFFFu - s);
    const uint32_t mask = (1u << bits) - 1;
    uint32_t slot = map_slot(key, bits);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        uint64_t cur = map_key[slot];
        if (cur == EMPTY_KEY) {
            cur = atomicCAS((unsigned long long *)&map_key[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (cur == EMPTY_KEY) cur = key;
        }
        if (cur == key) { atomicMax(&map_val[slot], val); return; }
        slot = (slot + 1) & mask;
    }
    *overflow = 1u;
}

__device__ __forceinline__ uint32_t map_centre(uint64_t key, const uint64_t *__restrict__ map_key, const unsigned long long *__restrict__ map_val, int bits)
{
    const uint32_t mask = (1u << bits) - 1;
    uint32_t slot = map_slot(key, bits);
    for (;;) {
        if (map_key[slot] == key) return 0xFFFFFFFFu - (uint32_t)map_val[slot];
        slot = (slot + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void lc_verify(const uint8_t *__restrict__ res, const uint64_t *__restrict__ off, uint32_t n, int m,
                                                 const uint64_t *__restrict__ sel_key, const uint32_t *__restrict__ sel_pos, const uint32_t *__restrict__ sel_cnt,
                                                 const uint64_t *__restrict__ map_key, const unsigned long long *__restrict__ map_val, int bits,
                                                 double min_id, double min_cov, uint32_t *__restrict__ acc, uint32_t *__restrict__ nacc,
                                                 unsigned long long *__restrict__ stats)
{
    const int lane = threadIdx.x & 63;
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n) return;
    const uint8_t *qs = res + off[s];
    const long long Ls = (long long)(off[s + 1] - off[s]);
    uint32_t seen_c[MAX_M], acc_local[MAX_M];        // wave-uniform bookkeeping (every lane holds the same values)
    long long seen_d[MAX_M];
    int ns = 0;
    uint32_t n_acc = 0, n_ver = 0;
    const uint32_t cnt = sel_cnt[s];
    for (uint32_t r = 0; r < cnt; ++r) {
        const uint64_t key = sel_key[(uint64_t)s * m + r];
        const uint32_t c = map_centre(key, map_key, map_val, bits);
        if (c == s) continue;
        long long pc = -1;
        const uint32_t cc = sel_cnt[c];
        for (uint32_t r2 = 0; r2 < cc; ++r2)
            if (sel_key[(uint64_t)c * m + r2] == key) { pc = sel_pos[(uint64_t)c * m + r2]; break; }
        const long long d = pc - (long long)sel_pos[(uint64_t)s * m + r];
        bool dup = false;
        for (int z = 0; z < ns; ++z) dup = dup || (seen_c[z] == c && seen_d[z] == d);
        if (dup) continue;
        seen_c[ns] = c; seen_d[ns] = d; ++ns;
        ++n_ver;
        const uint8_t *qc = res + off[c];
        const long long Lc = (long long)(off[c + 1] - off[c]);
        const long long x0 = d < 0 ? -d : 0, x1 = Ls < Lc - d ? Ls : Lc - d;
        long long match = 0;
        for (long long x = x0 + lane; x < x1; x += 64) match += (qs[x] == qc[x + d]) ? 1 : 0;
        for (int dd = 32; dd > 0; dd >>= 1) match += __shfl_xor(match, dd, 64);
        const long long ovl = x1 - x0;
        if (ovl > 0 && (double)match >= min_id * (double)ovl && (double)ovl >= min_cov * (double)Lc && (double)ovl >= min_cov * (double)Ls) {
            bool have = false;
            for (uint32_t z = 0; z < n_acc; ++z) have = have || (acc_local[z] == c);
            if (!have) acc_local[n_acc++] = c;
        }
    }
    if (lane == 0) {
        for (uint32_t z = 0; z < n_acc; ++z) acc[(uint64_t)s * m + z] = acc_local[z];
        nacc[s] = n_acc;
        if (n_ver) atomicAdd(&stats[0], (unsigned long long)n_ver);
        if (n_acc) atomicAdd(&stats[1], (unsigned long long)n_acc);
    }
}

// priority: longer first, then lower index
__device__ __forceinline__ bool higher(uint32_t la, uint32_t a, uint32_t lb, uint32_t b) { return la > lb || (la == lb && a < b); }

// status: 0 undecided, 1 representative, 2 member.  s becomes a member of its highest-priority accepted centre that is a
// representative once every accepted centre of higher priority is a member; a representative once all of them are members.
__global__ __launch_bounds__(256) void lc_assign(const uint64_t *__restrict__ off, uint32_t n, int m, const uint32_t *__restrict__ acc,
                                                 const uint32_t *__restrict__ nacc, uint32_t *status, uint32_t *__restrict__ rep, uint32_t *__restrict__ pending)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    if (__hip_atomic_load(&status[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const uint32_t k = nacc[s];
    uint32_t best = 0xFFFFFFFFu, best_len = 0;          // highest-priority centre that is a representative
    uint32_t blocker = 0xFFFFFFFFu, blocker_len = 0;    // highest-priority undecided centre
    for (uint32_t z = 0; z < k; ++z) {
        const uint32_t c = acc[(uint64_t)s * m + z];
        const uint32_t lc = (uint32_t)(off[c + 1] - off[c]);
        const uint32_t st = __hip_atomic_load(&status[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st == 1 && (best == 0xFFFFFFFFu || higher(lc, c, best_len, best))) { best = c; best_len = lc; }
        if (st == 0 && (blocker == 0xFFFFFFFFu || higher(lc, c, blocker_len, blocker))) { blocker = c; blocker_len = lc; }
    }
    if (best != 0xFFFFFFFFu && (blocker == 0xFFFFFFFFu || higher(best_len, best, blocker_len, blocker))) {
        rep[s] = best;
        __hip_atomic_store(&status[s], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (best == 0xFFFFFFFFu && blocker == 0xFFFFFFFFu) {
        rep[s] = s;
        __hip_atomic_store(&status[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *pending = 1u;
    }
}

}  // namespace

int pep_k9_linclust(pep_ctx *ctx, const uint8_t *h_res, const uint64_t *h_off, uint32_t n, int base, int k, int m, double min_id, double min_cov,
                    uint32_t *h_rep, uint64_t *h_stats)
{
    if (n == 0) return PEP_OK;
    if (m < 1 || m > MAX_M || k < 1 || k > 32 || base < 2) return pep_fail(ctx, PEP_ERR_ARG, "pep_linclust: invalid k / m / base");
    {
        double span = 1.;
        for (int i = 0; i < k; ++i) span *= base;
        if (span > 18446744073709551615.0 / 2) return pep_fail(ctx, PEP_ERR_ARG, "pep_linclust: k-mer value does not fit 63 bits");
    }
    const uint64_t total = h_off[n];
    hipStream_t st = ctx->stream;
    DevBuf *W = ctx->ws;
    PEP_TRY(dev_reserve(ctx, W[0], total + 64));
    PEP_TRY(dev_reserve(ctx, W[1], ((uint64_t)n + 1) * 8));
    PEP_TRY(dev_reserve(ctx, W[2], (uint64_t)n * m * 8 + 8));
    PEP_TRY(dev_reserve(ctx, W[3], (uint64_t)n * m * 4 + 8));
    PEP_TRY(dev_reserve(ctx, W[4], ((uint64_t)n + 1) * 4));
    int bits = 10;
    while ((1ull << bits) < 2ull * n * m) ++bits;
    if (bits > 31) return pep_fail(ctx, PEP_ERR_LIMIT, "pep_linclust: too many k-mers for the centre map");
    const uint64_t cap = 1ull << bits;
    PEP_TRY(dev_reserve(ctx, W[5], cap * 8));
    PEP_TRY(dev_reserve(ctx, W[6], cap * 8));
    PEP_TRY(dev_reserve(ctx, W[7], (uint64_t)n * m * 4 + 8));
    PEP_TRY(dev_reserve(ctx, W[8], ((uint64_t)n + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[9], ((uint64_t)n + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[10], ((uint64_t)n + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[11], 64));
    uint8_t *res = W[0].as<uint8_t>();
    uint64_t *off = W[1].as<uint64_t>(), *sel_key = W[2].as<uint64_t>(), *map_key = W[5].as<uint64_t>();
    uint32_t *sel_pos = W[3].as<uint32_t>(), *sel_cnt = W[4].as<uint32_t>(), *acc = W[7].as<uint32_t>(), *nacc = W[8].as<uint32_t>();
    uint32_t *status = W[9].as<uint32_t>(), *rep = W[10].as<uint32_t>();
    unsigned long long *map_val = W[6].as<unsigned long long>();
    uint32_t *flags = W[11].as<uint32_t>();
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(flags + 4);
    if (total) PEP_HIP(ctx, hipMemcpyAsync(res, h_res, total, hipMemcpyHostToDevice, st));
    PEP_HIP(ctx, hipMemcpyAsync(off, h_off, ((uint64_t)n + 1) * 8, hipMemcpyHostToDevice, st));
    PEP_HIP(ctx, hipMemsetAsync(map_key, 0xFF, cap * 8, st));
    PEP_HIP(ctx, hipMemsetAsync(map_val, 0, cap * 8, st));
    PEP_HIP(ctx, hipMemsetAsync(flags, 0, 64, st));
    PEP_HIP(ctx, hipMemsetAsync(status, 0, ((uint64_t)n + 1) * 4, st));
    const unsigned gw = (unsigned)ceil_div(n, 4);
    hipLaunchKernelGGL(lc_select, dim3(gw), dim3(256), 0, st, (const uint8_t *)res, (const uint64_t *)off, n, base, k, m, sel_key, sel_pos, sel_cnt);
    hipLaunchKernelGGL(lc_insert, dim3((unsigned)ceil_div((uint64_t)n * m, 256)), dim3(256), 0, st, (const uint64_t *)off, n, m, (const uint64_t *)sel_key,
                       (const uint32_t *)sel_cnt, map_key, map_val, bits, flags);
    hipLaunchKernelGGL(lc_verify, dim3(gw), dim3(256), 0, st, (const uint8_t *)res, (const uint64_t *)off, n, m, (const uint64_t *)sel_key, (const uint32_t *)sel_pos,
                       (const uint32_t *)sel_cnt, (const uint64_t *)map_key, (const unsigned long long *)map_val, bits, min_id, min_cov, acc, nacc, stats);
    PEP_HIP(ctx, hipGetLastError());
    for (int round = 0; round < 100000; ++round) {
        PEP_HIP(ctx, hipMemsetAsync(flags + 1, 0, 4, st));
        hipLaunchKernelGGL(lc_assign, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const uint64_t *)off, n, m, (const uint32_t *)acc, (const uint32_t *)nacc,
                           status, rep, flags + 1);
        uint32_t h_flags[2];
        PEP_HIP(ctx, hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, st));
        PEP_HIP(ctx, hipStreamSynchronize(st));
        if (h_flags[0]) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_linclust: centre map overflow");
        if (!h_flags[1]) break;
    }
    PEP_HIP(ctx, hipMemcpyAsync(h_rep, rep, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    std::vector<uint32_t> cnt(n);
    PEP_HIP(ctx, hipMemcpyAsync(cnt.data(), sel_cnt, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    unsigned long long hs[2];
    PEP_HIP(ctx, hipMemcpyAsync(hs, stats, 16, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipStreamSynchronize(st));
    if (h_stats) {
        uint64_t tot = 0;
        for (uint32_t i = 0; i < n; ++i) tot += cnt[i];
        h_stats[0] = tot; h_stats[1] = hs[0]; h_stats[2] = hs[1];
    }
    return PEP_OK;
}
