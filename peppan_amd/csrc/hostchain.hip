// Host-side C++ of RunBlast.run's numeric chain between the search and the caller (no GPU work, no context):
//   pep_table_from_hits   hit records of a search -> the columns of the reference's table rows
//                           translated tool: parseDiamond's coordinate algebra and filters        uberBlast.py:25-58
//                           nucleotide tool: parseBlast's columns and filters                     uberBlast.py:275-290, 311-320
//   pep_cols_fix_end      RunBlast.fixEnd over all rows                                           uberBlast.py:462-480
//   pep_cols_order        the sort that ends RunBlast.run: query, reference, score - stable        uberBlast.py:375
//   pep_cols_gather       rows picked from every column of a table in one pass
// These were a chain of numpy expressions - some forty passes over 70 000 rows, 16 ms of the 31 ms the reference's hot call
// (PEPPAN.py:229-230) took through the drop-in, and the per-genome bookkeeping of the mapping path.  Every float expression keeps the
// operand order of the numpy form (IEEE double, no contraction): golden G3 / G4 / G6 / G8 hold the results to the last bit.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <atomic>
#include <thread>
#include <vector>

namespace {

// Threads for one pass over `rows` rows of a table.  The passes below are a few hundred microseconds of one core each on the hot call's 70 000 rows and stand
// one behind the other on its critical path; threads are started per call - nothing outlives it, a forked child inherits nothing.
// PEPPAN_HOST_THREADS: the most a call may use (the Python side sets it from the CPUs the container grants; the mapping pool's workers get 1).
std::atomic<int> g_host_threads{0};             // 0: not decided yet

int host_threads(uint64_t rows)
{
    int granted = g_host_threads.load(std::memory_order_relaxed);
    if (granted <= 0) {
        const char *e = getenv("PEPPAN_HOST_THREADS");
        granted = e ? atoi(e) : 0;
        if (granted <= 0) granted = (int)std::min(4u, std::max(1u, std::thread::hardware_concurrency() / 4));
        granted = std::min(granted, 16);
        g_host_threads.store(granted, std::memory_order_relaxed);
    }
    if (rows < 16384) return 1;
    return (int)std::min<uint64_t>((uint64_t)granted, rows / 8192);
}

// f(t, lo, hi) for T even chunks of [0, n): chunk 0 on the calling thread
template <class F> void in_chunks(uint64_t n, int T, F f)
{
    if (T <= 1) { f(0, (uint64_t)0, n); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)T);
    for (int t = 1; t < T; ++t) {
        const uint64_t lo = n * (uint64_t)t / (uint64_t)T, hi = n * (uint64_t)(t + 1) / (uint64_t)T;
        try { pool.emplace_back([=]() { f(t, lo, hi); }); }
        catch (...) { f(t, lo, hi); }                    // (no thread to be had: the chunk runs here - nothing may leave this library as a C++ exception)
    }
    f(0, (uint64_t)0, n / (uint64_t)T);
    for (auto &th : pool) th.join();
}

// numpy.round(x, 3): multiply, round half to even, divide
inline double round3(double x) { return std::nearbyint(x * 1000.0) / 1000.0; }

// float('%.3f' % x) (blastn prints pident with three decimals, uberBlast.py:282): the correctly rounded 3-digit decimal of the double, read
// back.  round3 gives the same value except when x * 1000 sits within an ulp of a tie: those go through the decimal string.
inline double three_decimals(double v)
{
    const double scaled = v * 1000.0;
    const double frac = std::fabs(scaled - std::floor(scaled) - 0.5);
    if (frac < 1e-6) {
        char buf[64];
        snprintf(buf, sizeof buf, "%.3f", v);
        return strtod(buf, nullptr);
    }
    return round3(v);
}

}   // namespace

extern "C" {

// the most threads a pass of this file may use from now on (1 .. 16; 0: back to PEPPAN_HOST_THREADS / the default); returns the value in force before
int pep_set_host_threads(int n)
{
    host_threads(0);
    return g_host_threads.exchange(n <= 0 ? 0 : std::min(n, 16));
}

int64_t pep_table_from_hits(int32_t tool, uint64_t n, const pep_hit *hits, const uint32_t *cigar, uint64_t n_cigar, const pep_query_meta *q_meta,
                            const pep_target_meta *t_meta, const int64_t *q_len, const int64_t *r_len, const int64_t *t_seq, const uint8_t *t_rev,
                            const int64_t *win_off, const int64_t *home_lo, const int64_t *home_hi, const double *evalue, double min_id, double min_cov,
                            double min_ratio, pep_hit_cols *out, uint32_t *arena_out, const uint32_t *nt_match)
{
    if ((n && (!hits || !out || !q_len || !r_len)) || (n_cigar && (!cigar || !arena_out))) return PEP_ERR_ARG;
    if (tool == 0 && n && (!q_meta || !t_meta)) return PEP_ERR_ARG;
    if (tool == 1 && n && (!t_seq || !t_rev)) return PEP_ERR_ARG;
    if (tool != 0 && tool != 1) return PEP_ERR_ARG;
    // rows in chunks, a thread each: chunk t writes its rows from row lo_t on and the gaps the filters left are closed afterwards (few rows fail the cuts)
    const int T = host_threads(n);
    std::vector<int64_t> made((size_t)T, 0);
    const int64_t unit = tool == 0 ? 3 : 1;       // nucleotides per CIGAR column
    auto chunk = [&](int t, uint64_t lo, uint64_t hi) {
    // the CIGAR arena in nucleotide units: the translated tool's runs count residues
    const uint64_t c_lo = n_cigar * (uint64_t)t / (uint64_t)T, c_hi = n_cigar * (uint64_t)(t + 1) / (uint64_t)T;
    if (tool == 0) for (uint64_t k = c_lo; k < c_hi; ++k) arena_out[k] = ((cigar[k] >> 2) * 3u) << 2 | (cigar[k] & 3u);
    else if (arena_out != cigar) memcpy(arena_out + c_lo, cigar + c_lo, (c_hi - c_lo) * sizeof(uint32_t));
    int64_t m = (int64_t)lo;
    for (uint64_t i = lo; i < hi; ++i) {
        const pep_hit &h = hits[i];
        if (h.cigar_off + h.cigar_runs > n_cigar) { made[(size_t)t] = PEP_ERR_ARG; return; }
        int64_t gap_cols = 0, gap_open = 0, m_cols = 0, long_gap_cols = 0;      // (in the units of the runs: residues for the translated tool)
        for (uint32_t k = 0; k < h.cigar_runs; ++k) {
            const uint32_t run = cigar[h.cigar_off + k];
            if (run & 3u) { gap_cols += run >> 2; ++gap_open; if ((int64_t)(run >> 2) * unit > 3) long_gap_cols += run >> 2; }
            else m_cols += run >> 2;
        }
        if (tool == 0) {
            // parseDiamond: names q:frame / r:frame:offset, CIGAR x 3, identity from NM, coordinates back to nucleotides of either strand
            const pep_query_meta &qm = q_meta[h.q];
            const pep_target_meta &tm = t_meta[h.t];
            const int64_t qseq = qm.seq, qf = qm.frame, rseq = tm.seq, rf = tm.frame, rx = tm.chunk_off;
            const int64_t ql = q_len[qseq], rl = r_len[rseq];
            const int64_t qs_aa = h.q_start, rs_aa = (int64_t)h.t_start + rx;
            const int64_t qmatch = (int64_t)h.q_end - (int64_t)h.q_start + 1, rmatch = (int64_t)h.t_end - (int64_t)h.t_start + 1;
            const int64_t cl = 3 * (int64_t)h.aln_len;
            const double variation = 3.0 * (double)h.nm;
            const double iden = 1.0 - round3(variation / (double)cl);
            if (!((double)(qmatch * 3) >= min_cov && (double)qmatch * 3.0 / (double)ql >= min_ratio && iden >= min_id)) continue;
            const bool fwd = rf <= 3;
            out->qi[m] = qseq; out->ri[m] = rseq; out->iden[m] = iden; out->aln[m] = cl;
            out->mis[m] = (int64_t)(variation - (double)(3 * gap_cols)); out->gap[m] = gap_open;
            out->qs[m] = qs_aa * 3 + qf - 3; out->qe[m] = (qs_aa + qmatch - 1) * 3 + qf - 1;
            out->ss[m] = fwd ? rs_aa * 3 + rf - 3 : rl - (rs_aa * 3 + rf - 6) + 1;
            out->se[m] = fwd ? (rs_aa + rmatch - 1) * 3 + rf - 1 : rl - ((rs_aa + rmatch - 1) * 3 + rf - 4) + 1;
            out->evalue[m] = 0.0; out->score[m] = (double)h.score; out->ql[m] = ql; out->sl[m] = rl;
        } else {
            // parseBlast: the subject strand from the target, identity with blastn's three printed decimals, e-value as given
            const int64_t ri = t_seq[h.t];
            const bool rev = t_rev[h.t] != 0;
            const int64_t ql = q_len[h.q], sl = r_len[ri];
            const int64_t qs = h.q_start, qe = h.q_end;
            int64_t ts = h.t_start, te = h.t_end;
            if (win_off) {
                // targets that are windows of a long strand: back to strand coordinates; a hit belongs to the window whose home stretch holds its midpoint
                ts += win_off[h.t]; te += win_off[h.t];
                const int64_t mid = (ts + te - 2) / 2;           // (non-negative: floor division as in the numpy form)
                if (!(mid >= home_lo[h.t] && mid < home_hi[h.t])) continue;
            }
            const double iden = three_decimals(100.0 * (double)h.n_ident / (double)h.aln_len) / 100.0;
            const int64_t span = qe - qs + 1;
            if (!(iden >= min_id && (double)span >= min_cov && (double)span >= min_ratio * (double)ql)) continue;
            out->qi[m] = h.q; out->ri[m] = ri; out->iden[m] = iden; out->aln[m] = h.aln_len;
            out->mis[m] = (int64_t)h.aln_len - (int64_t)h.n_ident - gap_cols; out->gap[m] = gap_open;
            out->qs[m] = qs; out->qe[m] = qe;
            out->ss[m] = rev ? sl - ts + 1 : ts; out->se[m] = rev ? sl - te + 1 : te;
            out->evalue[m] = evalue ? evalue[i] : 0.0; out->score[m] = (double)h.score; out->ql[m] = ql; out->sl[m] = sl;
        }
        if (nt_match) {
            // reScore mode 1 (uberBlast.py:397-415, cigar2score :226-249) on the row that was just made: identity and score from K7's counts - identical columns from
            // the device (the search counted them for its own hits: pep_set_nt_match), the gap counts from the CIGAR - in the float64 arithmetic of the numpy form:
            // identity = matches / (matches + mismatches + gap bases - gap bases of gaps longer than 3), score = 3 matches - mismatches - 5 gaps - gap bases,
            // both rounded to three decimals half to even.  (The cuts above looked at the tool's own identity, as the reference's parsers do.)
            const int64_t n_match = nt_match[i], n_mis = m_cols * unit - n_match, b_gap = gap_cols * unit, m_gap = long_gap_cols * unit;
            out->iden[m] = round3((double)n_match / (double)(n_match + n_mis + b_gap - m_gap));
            out->score[m] = round3((double)(n_match * 3 - n_mis - gap_open * (6 - 1) - b_gap * 1));
        }
        out->c_off[m] = (int64_t)h.cigar_off; out->c_runs[m] = h.cigar_runs;
        if (out->rid) out->rid[m] = -1;
        ++m;
    }
    made[(size_t)t] = m - (int64_t)lo;
    };
    in_chunks(n, T, chunk);
    int64_t m = 0;
    for (int t = 0; t < T; ++t) {
        if (made[(size_t)t] < 0) return made[(size_t)t];
        const int64_t lo = (int64_t)(n * (uint64_t)t / (uint64_t)T);
        if (lo != m && made[(size_t)t]) {
            const size_t bytes = (size_t)made[(size_t)t] * 8;
            int64_t *const ints[] = {out->qi, out->ri, out->aln, out->mis, out->gap, out->qs, out->qe, out->ss, out->se, out->ql, out->sl, out->c_off, out->c_runs, out->rid};
            double *const reals[] = {out->iden, out->evalue, out->score};
            for (int64_t *c : ints) if (c) memmove(c + m, c + lo, bytes);
            for (double *c : reals) memmove(c + m, c + lo, bytes);
        }
        m += made[(size_t)t];
    }
    return m;
}

// RunBlast.fixEnd for all rows: an alignment is stretched over an unaligned query head of at most se_lim bases / tail of at most ee_lim bases, as
// far as the reference sequence allows; the first / last CIGAR run grows by the same amount whatever its operation is.  The rows' runs are
// copied into a private arena (rows may share runs, and so may the caller's table): arena_out takes sum(c_runs) words, c_off is rewritten.
// Returns the number of rows that changed, or PEP_ERR_ARG - also for a row without CIGAR runs that would have to be extended (the reference
// fails on cigar[0] there, uberBlast.py:468).
int64_t pep_cols_fix_end(uint64_t n, pep_hit_cols *c, const uint32_t *arena_in, uint64_t n_arena_in, uint32_t *arena_out, double se_lim, double ee_lim)
{
    if (n && (!c || !arena_out)) return PEP_ERR_ARG;
    // where every chunk's runs start in the private arena: the sum of the runs in front of it
    const int T = host_threads(n);
    std::vector<int64_t> first((size_t)T + 1, 0), changed((size_t)T, 0);
    for (int t = 0; t < T; ++t) {
        int64_t sum = 0;
        for (uint64_t i = n * (uint64_t)t / (uint64_t)T, hi = n * (uint64_t)(t + 1) / (uint64_t)T; i < hi; ++i) {
            if (c->c_runs[i] < 0) return PEP_ERR_ARG;
            sum += c->c_runs[i];
        }
        first[(size_t)t + 1] = first[(size_t)t] + sum;
    }
    in_chunks(n, T, [&](int t, uint64_t lo, uint64_t hi) {
        int64_t at = first[(size_t)t], ch = 0;
        for (uint64_t i = lo; i < hi; ++i) {
            const int64_t head = c->qs[i] - 1, tail = c->ql[i] - c->qe[i];
            const bool fwd = c->se[i] > c->ss[i];
            int64_t d = 0, e = 0;
            if (head > 0 && (double)head <= se_lim) d = fwd ? std::min(head, c->ss[i] - 1) : std::min(head, c->sl[i] - c->ss[i]);
            if (tail > 0 && (double)tail <= ee_lim) e = fwd ? std::min(tail, c->sl[i] - c->se[i]) : std::min(tail, c->se[i] - 1);
            const int64_t runs = c->c_runs[i], off = c->c_off[i];
            if (((d || e) && runs <= 0) || off < 0 || (uint64_t)(off + runs) > n_arena_in) { changed[(size_t)t] = PEP_ERR_ARG; return; }
            for (int64_t k = 0; k < runs; ++k) arena_out[at + k] = arena_in[off + k];           // (a handful of words: a call of memcpy costs more than the copy)
            if (d || e) {
                arena_out[at] += (uint32_t)(d << 2);
                arena_out[at + runs - 1] += (uint32_t)(e << 2);
                c->qs[i] -= d; c->ss[i] += fwd ? -d : d;
                c->qe[i] += e; c->se[i] += fwd ? e : -e;
                ++ch;
            }
            c->c_off[i] = at;
            at += runs;
        }
        changed[(size_t)t] = ch;
    });
    int64_t total = 0;
    for (int t = 0; t < T; ++t) { if (changed[(size_t)t] < 0) return changed[(size_t)t]; total += changed[(size_t)t]; }
    return total;
}

// order[k] = the row that comes k-th when the table is sorted by (q_code, r_code, score), stable - the multi-column sort that ends RunBlast.run
// with the names replaced by codes that sort like them.  Codes must be non-negative: two LSD radix passes per code as far as its range needs,
// then the rows of one (query, reference) pair - a handful - by score.
int pep_cols_order(uint64_t n, const int64_t *q_code, const int64_t *r_code, const double *score, int64_t *order)
{
    if (n && (!q_code || !r_code || !score || !order)) return PEP_ERR_ARG;
    int64_t q_max = 0, r_max = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (q_code[i] < 0 || r_code[i] < 0) return PEP_ERR_ARG;
        q_max = std::max(q_max, q_code[i]); r_max = std::max(r_max, r_code[i]);
    }
    int q_bits = 0, r_bits = 0;
    while (q_bits < 63 && (q_max >> q_bits) != 0) ++q_bits;
    while (r_bits < 63 && (r_max >> r_bits) != 0) ++r_bits;
    auto by_score = [&](int64_t *rows, uint64_t a, uint64_t b) {          // rows of one pair by score (ascending; equal scores keep their order)
        if (b - a == 2) { if (score[rows[a + 1]] < score[rows[a]]) std::swap(rows[a], rows[a + 1]); }
        else std::stable_sort(rows + a, rows + b, [&](int64_t x, int64_t y) { return score[x] < score[y]; });
    };
    if (q_bits + r_bits <= 32 && n < ((uint64_t)1 << 32)) {
        // the usual case (names of a gene set: 14 + 14 bits at 10 000 genes): records (query code, reference code, row) in one word, LSD radix passes of 11 bits
        // over the code bits - counting tables of 16 KiB that stay in the L1, rows read and written in order - instead of passes of 16 bits that gather the
        // codes through the row order (0.82 -> 0.3 ms at the hot call's 70 000 rows)
        const int key_bits = q_bits + r_bits, passes = (key_bits + 10) / 11;
        std::vector<uint64_t> rec_a((size_t)n), rec_b((size_t)n);
        std::vector<uint32_t> cnt((size_t)std::max(passes, 1) * 2048, 0);
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t key = ((uint64_t)q_code[i] << r_bits) | (uint64_t)r_code[i];
            rec_a[(size_t)i] = (key << 32) | i;
            for (int p = 0; p < passes; ++p) ++cnt[(size_t)p * 2048 + ((key >> (11 * p)) & 2047u)];
        }
        uint64_t *src = rec_a.data(), *dst = rec_b.data();
        for (int p = 0; p < passes; ++p) {
            uint32_t *c = cnt.data() + (size_t)p * 2048, run = 0;
            for (int b = 0; b < 2048; ++b) { const uint32_t k = c[b]; c[b] = run; run += k; }
            const int shift = 32 + 11 * p;
            for (uint64_t i = 0; i < n; ++i) dst[c[(src[i] >> shift) & 2047u]++] = src[i];
            std::swap(src, dst);
        }
        for (uint64_t i = 0; i < n; ++i) order[i] = (int64_t)(src[i] & 0xFFFFFFFFull);
        for (uint64_t a = 0; a < n;) {
            uint64_t b = a + 1;
            while (b < n && (src[b] >> 32) == (src[a] >> 32)) ++b;
            if (b - a > 1) by_score(order, a, b);
            a = b;
        }
        return PEP_OK;
    }
    std::vector<int64_t> tmp((size_t)n);
    for (uint64_t i = 0; i < n; ++i) order[i] = (int64_t)i;
    int64_t *src = order, *dst = tmp.data();
    std::vector<uint64_t> cnt(65537);
    auto pass = [&](const int64_t *code, int shift) {
        std::fill(cnt.begin(), cnt.end(), 0);
        for (uint64_t i = 0; i < n; ++i) ++cnt[(size_t)((code[src[i]] >> shift) & 0xFFFF) + 1];
        for (int b = 0; b < 65536; ++b) cnt[(size_t)b + 1] += cnt[(size_t)b];
        for (uint64_t i = 0; i < n; ++i) dst[cnt[(size_t)((code[src[i]] >> shift) & 0xFFFF)]++] = src[i];
        std::swap(src, dst);
    };
    for (int shift = 0; shift < 64 && (r_max >> shift) != 0; shift += 16) pass(r_code, shift);
    for (int shift = 0; shift < 64 && (q_max >> shift) != 0; shift += 16) pass(q_code, shift);
    for (uint64_t a = 0; a < n;) {
        uint64_t b = a + 1;
        while (b < n && q_code[src[b]] == q_code[src[a]] && r_code[src[b]] == r_code[src[a]]) ++b;
        if (b - a > 1) by_score(src, a, b);
        a = b;
    }
    if (src != order) memcpy(order, src, (size_t)n * sizeof(int64_t));
    return PEP_OK;
}

// order = numpy.lexsort(keys): the stable order by the LAST key, ties by the one before it, ... - for int64 keys of modest range (codes, coordinates), LSD radix passes of
// 11 bits over (key - its minimum), least significant key first.  The -f and -m fronts of the mapping path sort every genome's table by four such keys (numpy's lexsort
// is four merge sorts through an index: 1 ms per 12 000 rows; this is ~0.15).  PEP_ERR_LIMIT when a key's range needs more than 44 bits (the caller takes numpy's).
int pep_lex_order(uint64_t n, int32_t n_keys, const int64_t *const *keys, int64_t *order)
{
    if (n_keys < 1 || !keys || (n && !order)) return PEP_ERR_ARG;
    if (n >= ((uint64_t)1 << 32)) return PEP_ERR_LIMIT;
    std::vector<uint32_t> a((size_t)n), b((size_t)n);
    for (uint64_t i = 0; i < n; ++i) a[(size_t)i] = (uint32_t)i;
    uint32_t *src = a.data(), *dst = b.data();
    std::vector<uint64_t> val((size_t)n);
    for (int32_t k = 0; k < n_keys; ++k) {
        const int64_t *key = keys[k];
        if (!key && n) return PEP_ERR_ARG;
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (uint64_t i = 0; i < n; ++i) { lo = std::min(lo, key[i]); hi = std::max(hi, key[i]); }
        if (n == 0 || lo == hi) continue;
        const uint64_t range = (uint64_t)hi - (uint64_t)lo;
        int bits = 0;
        while (bits < 64 && (range >> bits) != 0) ++bits;
        if (bits > 44) return PEP_ERR_LIMIT;
        for (uint64_t i = 0; i < n; ++i) val[(size_t)i] = (uint64_t)key[src[i]] - (uint64_t)lo;       // the key in the current order: the passes below move it along with the row
        // (value and row in one word would need 44 + 32 bits: the values are carried in a second array of the same order instead)
        std::vector<uint64_t> val2((size_t)n);
        uint64_t *vs = val.data(), *vd = val2.data();
        for (int shift = 0; shift < bits; shift += 11) {
            uint32_t cnt[2048] = {0};
            for (uint64_t i = 0; i < n; ++i) ++cnt[(vs[i] >> shift) & 2047u];
            uint32_t run = 0;
            for (int x = 0; x < 2048; ++x) { const uint32_t c = cnt[x]; cnt[x] = run; run += c; }
            for (uint64_t i = 0; i < n; ++i) { const uint32_t at = cnt[(vs[i] >> shift) & 2047u]++; dst[at] = src[i]; vd[at] = vs[i]; }
            std::swap(src, dst); std::swap(vs, vd);
        }
    }
    for (uint64_t i = 0; i < n; ++i) order[i] = (int64_t)src[i];
    return PEP_OK;
}

// dst[c][k] = src[c][idx[k]] for n_cols columns of 8-byte elements (every column of a hit table is int64 or double): HitTable.take in one call
int pep_cols_gather(int32_t n_cols, const void *const *src, void *const *dst, const int64_t *idx, uint64_t n_idx, uint64_t n_src)
{
    if (n_cols < 0 || (n_cols && (!src || !dst)) || (n_idx && !idx)) return PEP_ERR_ARG;
    for (uint64_t k = 0; k < n_idx; ++k)
        if (idx[k] < 0 || (uint64_t)idx[k] >= n_src) return PEP_ERR_ARG;
    // the columns dealt to the threads (a column's gather is one stream of writes and reads inside one array)
    const int T = std::min<int>(host_threads(n_idx), std::max(1, (int)n_cols));
    in_chunks((uint64_t)n_cols, T, [&](int, uint64_t lo, uint64_t hi) {
        for (uint64_t c = lo; c < hi; ++c) {
            const uint64_t *s = static_cast<const uint64_t *>(src[c]);
            uint64_t *d = static_cast<uint64_t *>(dst[c]);
            for (uint64_t k = 0; k < n_idx; ++k) d[k] = s[idx[k]];
        }
    });
    return PEP_OK;
}

}   // extern "C"


// np.argsort(x.astype(object)) for float64 x without NaN: numpy sorts an object column with its generic index quicksort (npy_aquicksort: median of
// three, partitions of more than 16 elements, the larger part pushed, insertion sort below; heapsort once the depth limit 2 * floor(log2 n) is used up)
// driven by the elements' own comparison - for Python floats the comparison of the doubles.  The order it leaves EQUAL elements in depends on every
// swap it makes, so the same steps are taken here on the doubles (the .tab store's rows of equal score come in that order: StoreBlock, mapbsn.py -
// 3 ms per genome as 170 000 comparisons of Python objects).  Returns 0, or PEP_ERR_LIMIT when the depth limit is reached (the caller then asks numpy itself).
extern "C" int pep_argsort_object_order(const double *v, int64_t n, int64_t *tosort)
{
    if (n < 0 || (n && (!v || !tosort))) return PEP_ERR_ARG;
    for (int64_t i = 0; i < n; ++i) tosort[i] = i;
    if (n < 2) return PEP_OK;
    auto lt = [&](int64_t a, int64_t b) { return v[a] < v[b]; };           // (compare(a, b) < 0)
    int64_t *pl = tosort, *pr = tosort + n - 1;
    int64_t *stack[128], **sptr = stack;
    int depth[128], *psdepth = depth;
    int msb = 0;
    for (uint64_t k = (uint64_t)n; k >>= 1;) ++msb;
    int cdepth = msb * 2;
    for (;;) {
        if (cdepth < 0) return PEP_ERR_LIMIT;
        while (pr - pl > 15) {                                  // (SMALL_QUICKSORT of the generic sort: partitions of 16 and fewer go to the insertion sort)
            int64_t *pm = pl + ((pr - pl) >> 1);
            if (lt(*pm, *pl)) std::swap(*pm, *pl);
            if (lt(*pr, *pm)) std::swap(*pr, *pm);
            if (lt(*pm, *pl)) std::swap(*pm, *pl);
            const int64_t vp = *pm;
            int64_t *pi = pl, *pj = pr - 1;
            std::swap(*pm, *pj);
            for (;;) {
                do { ++pi; } while (lt(*pi, vp) && pi < pj);          // (the generic sort guards its scans: a user type's comparison need not be consistent)
                do { --pj; } while (lt(vp, *pj) && pi < pj);
                if (pi >= pj) break;
                std::swap(*pi, *pj);
            }
            int64_t *pk = pr - 1;
            std::swap(*pi, *pk);
            if (pi - pl < pr - pi) { *sptr++ = pi + 1; *sptr++ = pr; pr = pi - 1; }
            else { *sptr++ = pl; *sptr++ = pi - 1; pl = pi + 1; }
            *psdepth++ = --cdepth;
        }
        for (int64_t *pi = pl + 1; pi <= pr; ++pi) {
            const int64_t vi = *pi;
            int64_t *pj = pi, *pk = pi - 1;
            while (pj > pl && lt(vi, *pk)) *pj-- = *pk--;
            *pj = vi;
        }
        if (sptr == stack) break;
        pr = *(--sptr);
        pl = *(--sptr);
        cdepth = *(--psdepth);
    }
    return PEP_OK;
}
