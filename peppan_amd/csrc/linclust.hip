// K9: linear-time clustering that replaces `mmseqs createdb / linclust / createtsv` (clust.py:62-66).
// Algorithm = oracle_linclust (oracle/align_oracle.c): min-hash k-mer selection, centre = longest sequence
// per k-mer, verification of each member against its centres (ungapped on the k-mer diagonal first; pairs that fail go through
// the banded Smith-Waterman engine of the search, K5/K6, in the band that holds the diagonal), greedy assignment by priority.
//
//   lc_select  one wavefront per sequence: m rounds of "smallest (hash, pos) above the previous pick"      (HBM: L bytes read m times from L1/L2)
//   lc_insert  one thread per selected k-mer: open-addressing map key -> atomicMax(len << 32 | ~idx)       (random 16 B)
//   lc_verify  one wavefront per sequence: centre look-up, diagonal, lanes stride over the overlap          (HBM: 2 x overlap bytes per pair)
//   lc_assign  rounds of a monotone fixed point that equals the sequential greedy assignment                (random 4 B)
// All integer except the two threshold comparisons, done in IEEE double exactly as the oracle does.
#include "common.h"
#include <algorithm>

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
    // power-of-two alphabets (nucleotides, the pipeline's default): every lane scans a CONTIGUOUS stretch of positions and rolls the
    // k-mer value from one position to the next (one byte load, a shift and a multiply-add) instead of re-reading k bytes per position
    // and round - 17 byte loads per position, twenty times over, was nearly all of this kernel's time
    const bool pow2 = (base & (base - 1)) == 0;
    int shift = 0;
    while ((1 << shift) < base) ++shift;
    uint64_t top = 1;                                            // base^(k-1): weight of the letter that enters the window
    for (int i = 1; i < k; ++i) top *= (uint64_t)base;
    const uint32_t npos = L >= (uint32_t)k ? L - (uint32_t)k + 1 : 0;
    const uint32_t per = (npos + 63) / 64, p0 = (uint32_t)lane * per, p1 = min(npos, p0 + per);
    for (int r = 0; r < m; ++r) {
        uint64_t bh = ~0ull, bkey = 0;
        uint32_t bp = 0xFFFFFFFFu;
        if (pow2) {
            if (p0 < p1) {
                uint64_t key = 0;
                long long last_bad = -1;                          // last position holding a letter outside the alphabet
                for (int i = 0; i < k; ++i) {
                    const uint8_t c = q[p0 + i];
                    if (c >= base) last_bad = (long long)p0 + i; else key |= (uint64_t)c << (shift * i);
                }
                for (uint32_t p = p0; p < p1; ++p) {
                    if (last_bad < (long long)p) {
                        const uint64_t h = lc_mix(key);
                        if ((first || hp_less(last_h, last_p, h, p)) && hp_less(h, p, bh, bp)) { bh = h; bp = p; bkey = key; }
                    }
                    if (p + 1 < p1) {
                        const uint8_t c = q[p + k];
                        key >>= shift;
                        if (c >= base) last_bad = (long long)p + k; else key += top * c;
                    }
                }
            }
        } else
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
                                                 unsigned long long *__restrict__ stats, unsigned long long *__restrict__ gap_pair, int32_t *__restrict__ gap_bin,
                                                 unsigned long long *__restrict__ n_gap)
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
        } else if (lane == 0) {
            // not on this diagonal without gaps: hand (member, centre, band of the diagonal) to the gapped verification
            const unsigned long long g = atomicAdd(n_gap, 1ull);
            gap_pair[g] = ((unsigned long long)s << 32) | c;
            gap_bin[g] = (int32_t)((d + (1ll << 23)) >> 6);
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

// ---- gapped verification (host side): the alignment engine of the search run on (member, centre) pairs
// Scoring of the clustering alphabet: base 4 = nucleotides, +2 / -3, gap 6 + 2k (the blastn-like engine parameters); otherwise
// amino acids in the order ACDEFGHIKLMNPQRSTVWY, BLOSUM62, gap 11 + k; one extra code (base) = unknown residue.
void k9_params(int base, const pep_search_params &defaults, pep_search_params &P)
{
    P = defaults;
    for (int a = 0; a < 32; ++a)
        for (int b = 0; b < 32; ++b) P.sub[a * 32 + b] = -64;
    if (base == 4) {
        for (int a = 0; a < 5; ++a)
            for (int b = 0; b < 5; ++b) P.sub[a * 32 + b] = (int8_t)((a == b && a < 4) ? 2 : -3);
        P.gap_open = 6; P.gap_ext = 2;
    } else {
        static const char *letters = "ACDEFGHIKLMNPQRSTVWYX";
        for (int a = 0; a <= 20 && a <= base; ++a)
            for (int b = 0; b <= 20 && b <= base; ++b) P.sub[a * 32 + b] = defaults.sub[(letters[a] - 'A') * 32 + (letters[b] - 'A')];
        P.gap_open = 11; P.gap_ext = 1;
    }
    P.min_id_pct = 0.; P.min_qcov_pct = 0.;
    P.top_k = 0x7fffffff; P.n_splits = 1; P.hsp_mode = 0;
    P.reserved[0] = P.reserved[1] = P.reserved[2] = 0;
}

struct GapCand { uint32_t s, c; int32_t bin; };

// (member, centre) pairs accepted by the gapped alignment of the best band among `cands` (sorted by centre, member, bin; unique)
int k9_gapped(pep_ctx *ctx, const uint8_t *h_res, const uint64_t *h_off, int base, double min_id, double min_cov, const std::vector<GapCand> &cands,
              std::vector<std::pair<uint32_t, uint32_t>> &accepted)
{
    pep_search_params saved = ctx->params, defaults;
    pep_default_params(&defaults);
    k9_params(base, defaults, ctx->params);
    int rc = pep_upload_sub(ctx);
    // the sequence sets of the context are taken over: a later pep_search needs its inputs again (nucleotide inputs are re-translated)
    ctx->q_ready = ctx->t_ready = false;
    ctx->resid_from_nucl = false;
    ctx->t_class_ready = false;
    const uint64_t BYTES = 200ull << 20;                 // packed bytes per side and batch
    size_t at = 0;
    std::vector<uint8_t> codes;
    std::vector<uint64_t> off;
    std::vector<uint32_t> q_list, t_list;
    std::vector<uint64_t> keys;
    while (rc == PEP_OK && at < cands.size()) {
        // a batch = a run of whole centres (candidates are sorted by centre) within the size limits of one packed set
        q_list.clear(); t_list.clear(); keys.clear();
        std::vector<std::pair<uint32_t, uint32_t>> q_sorted;      // (member, local index) built after the batch is delimited
        uint64_t t_bytes = 0, q_bytes = 0;
        size_t end = at;
        while (end < cands.size()) {
            const uint32_t c = cands[end].c;
            size_t e2 = end;
            uint64_t add_q = 0;
            while (e2 < cands.size() && cands[e2].c == c) { add_q += (h_off[cands[e2].s + 1] - h_off[cands[e2].s]) + 32; ++e2; }
            const uint64_t add_t = (h_off[c + 1] - h_off[c]) + 32;
            if (end > at && (t_bytes + add_t > BYTES || q_bytes + add_q > BYTES || t_list.size() + 1 > (1u << 24) || (e2 - at) > (1u << 20))) break;
            t_bytes += add_t; q_bytes += add_q;
            t_list.push_back(c);
            end = e2;
        }
        // members of the batch, each once
        for (size_t i = at; i < end; ++i) q_list.push_back(cands[i].s);
        std::sort(q_list.begin(), q_list.end());
        q_list.erase(std::unique(q_list.begin(), q_list.end()), q_list.end());
        if (q_list.size() > PEP_MAX_QUERIES) { rc = pep_fail(ctx, PEP_ERR_LIMIT, "pep_linclust: gapped batch exceeds the query limit"); break; }    // (the parameters are restored below)
        auto upload = [&](SeqSet &set, const std::vector<uint32_t> &list, uint32_t max_n) {
            codes.clear(); off.assign(1, 0);
            for (uint32_t x : list) {
                codes.insert(codes.end(), h_res + h_off[x], h_res + h_off[x + 1]);
                off.push_back(codes.size());
            }
            if (codes.empty()) codes.push_back(0);
            return pep_upload_codes(ctx, set, codes.data(), off.data(), (uint32_t)list.size(), max_n);
        };
        rc = upload(ctx->q, q_list, PEP_MAX_QUERIES);
        if (rc == PEP_OK) rc = upload(ctx->t, t_list, PEP_MAX_TARGETS);
        if (rc != PEP_OK) break;
        for (size_t i = at, ti = 0; i < end; ++i) {
            while (t_list[ti] != cands[i].c) ++ti;
            const uint32_t qi = (uint32_t)(std::lower_bound(q_list.begin(), q_list.end(), cands[i].s) - q_list.begin());
            keys.push_back(((uint64_t)qi << 43) | ((uint64_t)ti << 18) | (uint64_t)(uint32_t)cands[i].bin);
        }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        rc = dev_reserve(ctx, ctx->ws[4], (keys.size() + 1) * 8);
        if (rc != PEP_OK) break;
        if (hipMemcpy(ctx->ws[4].p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { rc = pep_fail(ctx, PEP_ERR_HIP, "pep_linclust: key upload failed"); break; }
        std::vector<int32_t> min_score(q_list.size() + 1, 1);
        pep_result res;
        res.ctx = ctx;
        pep_materialise_staged(ctx);
        rc = pep_extend(ctx, ctx->ws[4].as<const uint64_t>(), keys.size(), min_score.data(), &res);
        if (rc != PEP_OK) break;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = pep_fail(ctx, PEP_ERR_HIP, "pep_linclust: stream sync failed"); break; }
        const pep_hit *hits = res.st_hits ? res.st_hits : res.hits.data();
        for (uint64_t h = 0; h < res.n_hits; ++h) {
            const pep_hit &x = hits[h];
            const uint32_t sidx = q_list[x.q], cidx = t_list[x.t];
            const double Ls = (double)(h_off[sidx + 1] - h_off[sidx]), Lc = (double)(h_off[cidx + 1] - h_off[cidx]);
            const double qspan = (double)(x.q_end - x.q_start + 1), tspan = (double)(x.t_end - x.t_start + 1);
            if ((double)x.n_ident >= min_id * (double)x.aln_len && qspan >= min_cov * Ls && tspan >= min_cov * Lc) accepted.emplace_back(sidx, cidx);
        }
        at = end;
    }
    ctx->params = saved;
    ctx->sub_ready = false;
    return rc;
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
    PEP_TRY(dev_reserve(ctx, W[12], (uint64_t)n * m * 8 + 8));
    PEP_TRY(dev_reserve(ctx, W[13], (uint64_t)n * m * 4 + 8));
    uint8_t *res = W[0].as<uint8_t>();
    uint64_t *off = W[1].as<uint64_t>(), *sel_key = W[2].as<uint64_t>(), *map_key = W[5].as<uint64_t>();
    uint32_t *sel_pos = W[3].as<uint32_t>(), *sel_cnt = W[4].as<uint32_t>(), *acc = W[7].as<uint32_t>(), *nacc = W[8].as<uint32_t>();
    uint32_t *status = W[9].as<uint32_t>(), *rep = W[10].as<uint32_t>();
    unsigned long long *map_val = W[6].as<unsigned long long>();
    uint32_t *flags = W[11].as<uint32_t>();
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(flags + 4);          // [0] verified, [1] accepted without gaps, [2] pairs for the gapped stage
    unsigned long long *gap_pair = W[12].as<unsigned long long>();
    int32_t *gap_bin = W[13].as<int32_t>();
    if (total) PEP_HIP(ctx, hipMemcpyAsync(res, h_res, total, hipMemcpyHostToDevice, st));
    PEP_HIP(ctx, hipMemcpyAsync(off, h_off, ((uint64_t)n + 1) * 8, hipMemcpyHostToDevice, st));
    PEP_HIP(ctx, hipMemsetAsync(map_key, 0xFF, cap * 8, st));
    PEP_HIP(ctx, hipMemsetAsync(map_val, 0, cap * 8, st));
    PEP_HIP(ctx, hipMemsetAsync(flags, 0, 64, st));
    const unsigned gw = (unsigned)ceil_div(n, 4);
    hipLaunchKernelGGL(lc_select, dim3(gw), dim3(256), 0, st, (const uint8_t *)res, (const uint64_t *)off, n, base, k, m, sel_key, sel_pos, sel_cnt);
    hipLaunchKernelGGL(lc_insert, dim3((unsigned)ceil_div((uint64_t)n * m, 256)), dim3(256), 0, st, (const uint64_t *)off, n, m, (const uint64_t *)sel_key,
                       (const uint32_t *)sel_cnt, map_key, map_val, bits, flags);
    hipLaunchKernelGGL(lc_verify, dim3(gw), dim3(256), 0, st, (const uint8_t *)res, (const uint64_t *)off, n, m, (const uint64_t *)sel_key, (const uint32_t *)sel_pos,
                       (const uint32_t *)sel_cnt, (const uint64_t *)map_key, (const unsigned long long *)map_val, bits, min_id, min_cov, acc, nacc, stats,
                       gap_pair, gap_bin, stats + 2);
    PEP_HIP(ctx, hipGetLastError());
    // ---- everything the later stages need comes to the host: the gapped stage reuses the workspaces of the alignment engine
    std::vector<uint32_t> cnt(n), h_nacc(n), h_acc((size_t)n * m);
    unsigned long long hs[3] = {0, 0, 0};
    uint32_t h_flags[2] = {0, 0};
    PEP_HIP(ctx, hipMemcpyAsync(cnt.data(), sel_cnt, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipMemcpyAsync(h_nacc.data(), nacc, (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipMemcpyAsync(h_acc.data(), acc, (uint64_t)n * m * 4, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipMemcpyAsync(hs, stats, 24, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, st));
    PEP_HIP(ctx, hipStreamSynchronize(st));
    if (h_flags[0]) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_linclust: centre map overflow");
    uint64_t n_accepted = hs[1];
    if (hs[2]) {
        std::vector<unsigned long long> gp(hs[2]);
        std::vector<int32_t> gb(hs[2]);
        PEP_HIP(ctx, hipMemcpy(gp.data(), gap_pair, hs[2] * 8, hipMemcpyDeviceToHost));
        PEP_HIP(ctx, hipMemcpy(gb.data(), gap_bin, hs[2] * 4, hipMemcpyDeviceToHost));
        std::vector<GapCand> cands(hs[2]);
        for (uint64_t i = 0; i < hs[2]; ++i) cands[i] = GapCand{(uint32_t)(gp[i] >> 32), (uint32_t)gp[i], gb[i]};
        std::sort(cands.begin(), cands.end(), [](const GapCand &a, const GapCand &b) { return a.c != b.c ? a.c < b.c : a.s != b.s ? a.s < b.s : a.bin < b.bin; });
        cands.erase(std::unique(cands.begin(), cands.end(), [](const GapCand &a, const GapCand &b) { return a.c == b.c && a.s == b.s && a.bin == b.bin; }), cands.end());
        std::vector<std::pair<uint32_t, uint32_t>> accepted;
        PEP_TRY(k9_gapped(ctx, h_res, h_off, base, min_id, min_cov, cands, accepted));
        for (const auto &sc : accepted) {
            uint32_t *list = h_acc.data() + (size_t)sc.first * m;
            bool have = false;
            for (uint32_t z = 0; z < h_nacc[sc.first]; ++z) have = have || (list[z] == sc.second);
            if (!have && h_nacc[sc.first] < (uint32_t)m) { list[h_nacc[sc.first]++] = sc.second; ++n_accepted; }
        }
        PEP_HIP(ctx, hipMemcpyAsync(off, h_off, ((uint64_t)n + 1) * 8, hipMemcpyHostToDevice, st));
        PEP_TRY(dev_reserve(ctx, W[7], (uint64_t)n * m * 4 + 8));
        PEP_TRY(dev_reserve(ctx, W[8], ((uint64_t)n + 1) * 4));
        PEP_TRY(dev_reserve(ctx, W[9], ((uint64_t)n + 1) * 4));
        PEP_TRY(dev_reserve(ctx, W[10], ((uint64_t)n + 1) * 4));
        PEP_TRY(dev_reserve(ctx, W[11], 64));
        acc = W[7].as<uint32_t>(); nacc = W[8].as<uint32_t>(); status = W[9].as<uint32_t>(); rep = W[10].as<uint32_t>(); flags = W[11].as<uint32_t>();
        PEP_HIP(ctx, hipMemcpyAsync(acc, h_acc.data(), (uint64_t)n * m * 4, hipMemcpyHostToDevice, st));
        PEP_HIP(ctx, hipMemcpyAsync(nacc, h_nacc.data(), (uint64_t)n * 4, hipMemcpyHostToDevice, st));
        PEP_HIP(ctx, hipMemsetAsync(flags, 0, 64, st));
    }
    PEP_HIP(ctx, hipMemsetAsync(status, 0, ((uint64_t)n + 1) * 4, st));
    for (int round = 0; round < 100000; ++round) {
        PEP_HIP(ctx, hipMemsetAsync(flags + 1, 0, 4, st));
        hipLaunchKernelGGL(lc_assign, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const uint64_t *)off, n, m, (const uint32_t *)acc, (const uint32_t *)nacc,
                           status, rep, flags + 1);
        PEP_HIP(ctx, hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, st));
        PEP_HIP(ctx, hipStreamSynchronize(st));
        if (!h_flags[1]) break;
    }
    PEP_TRY(pep_d2h_queue(ctx, h_rep, rep, (uint64_t)n * 4));
    PEP_HIP(ctx, hipStreamSynchronize(st));
    pep_d2h_finish(ctx);
    if (h_stats) {
        uint64_t tot = 0;
        for (uint32_t i = 0; i < n; ++i) tot += cnt[i];
        h_stats[0] = tot; h_stats[1] = hs[0]; h_stats[2] = n_accepted;
    }
    return PEP_OK;
}
