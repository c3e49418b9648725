// Host-side C++ of the order-dependent post-filters of the genes->genomes mapping (flags -f and -m of uberBlast):
//   pep_ovl_filter     RunBlast.ovlFilter                      uberBlast.py:417-452
//   pep_linear_merge   RunBlast.linearMerge + _linearMerge     uberBlast.py:100-218, 453-460
// Both are greedy passes whose result depends on the visiting order inside one (query, reference) neighbourhood, a few
// rows at a time - no data parallelism worth a kernel, but 40 us per row in Python.  They work on the numeric columns of
// the table, already sorted the way the reference sorts it (the Python wrapper does the sort and rebuilds the rows).
// Every float expression keeps the reference's operand order (double arithmetic, no contraction).
#include "common.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <set>

namespace {

struct Row {
    int64_t q, r, qs, qe, ss, se, ql, sl, rid;
    double iden, score;
};

struct Group {
    double score, iden;
    int64_t span;
    int flag;
    std::vector<int64_t> ids;      // local row indices: [i] or [i, j] or a joined chain
};

// Python list comparison a > b of [score, iden, span, flag, *ids]
bool group_greater(const Group &a, const Group &b)
{
    if (a.score != b.score) return a.score > b.score;
    if (a.iden != b.iden) return a.iden > b.iden;
    if (a.span != b.span) return a.span > b.span;
    if (a.flag != b.flag) return a.flag > b.flag;
    const size_t n = std::min(a.ids.size(), b.ids.size());
    for (size_t k = 0; k < n; ++k)
        if (a.ids[k] != b.ids[k]) return a.ids[k] > b.ids[k];
    return a.ids.size() > b.ids.size();
}

inline int64_t iabs(int64_t x) { return x < 0 ? -x : x; }

void pair_score(const Row &m1, const Row &m2, int64_t span1, int64_t span2, int64_t o_hi, int64_t o_lo, double *score, double *iden)
{
    if (o_hi > 0) {
        *score = m1.score + m2.score - (double)o_hi * std::min(m1.score / (double)span1, m2.score / (double)span2);
        *iden = (m1.iden * (double)span1 + m2.iden * (double)span2 - (double)o_hi * std::min(m1.iden, m2.iden)) / (double)(span1 + span2 - o_hi);
    } else {
        *score = m1.score + m2.score;
        *iden = (m1.iden * (double)span1 + m2.iden * (double)span2) / (double)(span1 + span2);
    }
    if (o_lo < 0) *score += (double)o_lo / 3.;
}

struct MergeOut {
    std::vector<int64_t> keep_seq;          // local indices in the insertion order of the reference's `used` dict (value 1 only)
    bool ascending = false;                 // no chaining happened: every row kept, in order
    std::vector<Group> assign;              // groups with iden >= 0, in `chosen` order
};

void merge_one_query(const Row *m, int64_t n, double gap_dist, double len_diff, MergeOut &out)
{
    const int64_t tail = 20;
    std::vector<Group> groups;
    groups.reserve((size_t)n + 4);
    std::vector<int64_t> head_edge, tail_edge;
    for (int64_t i = 0; i < n; ++i) {
        const Row &m1 = m[i];
        const int64_t span1 = m1.qe - m1.qs + 1;
        groups.push_back(Group{m1.score, m1.iden, span1, 0, {i}});
        if (m1.qs > tail && ((m1.ss > 0 && (double)(m1.ss - 1) <= gap_dist) || (m1.ss < 0 && (double)(m1.sl + m1.ss) < gap_dist))) tail_edge.push_back(i);
        if (m1.qe <= m1.ql - tail) {
            if ((m1.ss > 0 && (double)(m1.sl - m1.se) <= gap_dist) || (m1.ss < 0 && (double)(-1 - m1.se) < gap_dist)) head_edge.push_back(i);
            for (int64_t j = i + 1; j < n; ++j) {
                const Row &m2 = m[j];
                if (m1.r != m2.r || (m1.ss < 0 && m2.ss > 0) || (double)(m2.ss - m1.se - 1) >= gap_dist) break;
                const int64_t q_span = m2.qe - m1.qs + 1, r_span = m2.se - m1.ss + 1;
                if (std::fabs(m1.iden - m2.iden) > 0.3 || m1.ss + 3 >= m2.ss || m1.se + 3 >= m2.se || m1.qs + 3 >= m2.qs || m1.qe + 3 >= m2.qe ||
                    (double)(m2.qs - m1.qe - 1) >= gap_dist || (double)std::min(q_span, r_span) * len_diff < (double)std::max(q_span, r_span))
                    continue;
                const int64_t span2 = m2.qe - m2.qs + 1;
                const int64_t o1 = m1.qe - m2.qs + 1, o2 = m1.se - m2.ss + 1;
                double score, iden;
                pair_score(m1, m2, span1, span2, std::max(o1, o2), std::min(o1, o2), &score, &iden);
                if (score > m1.score && score > m2.score) groups.push_back(Group{score, iden, q_span, 0, {i, j}});
            }
        }
    }
    if (!head_edge.empty() && !tail_edge.empty()) {
        for (int64_t i : head_edge) {
            const Row &m1 = m[i];
            for (int64_t j : tail_edge) {
                const Row &m2 = m[j];
                if ((m1.r == m2.r && std::max(iabs(m1.ss), iabs(m1.se)) > std::min(iabs(m2.ss), iabs(m2.se))) || std::fabs(m1.iden - m2.iden) > 0.3 ||
                    m1.qs >= m2.qs || m1.qe >= m2.qe || (double)(m2.qs - m1.qe - 1) >= gap_dist)
                    continue;
                const int64_t q_span = m2.qe - m1.qs + 1;
                const int64_t g1 = m1.se < 0 ? -m1.se - 1 : m1.sl - m1.se;
                const int64_t g2 = m2.ss > 0 ? m2.ss - 1 : m2.sl + m2.ss;
                const int64_t r_span = m1.se - m1.ss + 1 + m2.se - m2.ss + 1 + g1 + g2;
                if ((double)(g1 + g2) >= gap_dist || (double)std::min(q_span, r_span) * len_diff < (double)std::max(q_span, r_span)) continue;
                const int64_t o1 = m1.qe - m2.qs + 1, o2 = -g1 - g2;
                double score, iden;
                pair_score(m1, m2, m1.qe - m1.qs + 1, m2.qe - m2.qs + 1, std::max(o1, o2), std::min(o1, o2), &score, &iden);
                if (score > m1.score && score > m2.score) groups.push_back(Group{score, iden, q_span, 1, {i, j}});
            }
        }
    }
    if ((int64_t)groups.size() <= n) {                // nothing to chain: every hit is its own group
        out.ascending = true;
        out.assign = std::move(groups);
        return;
    }
    std::stable_sort(groups.begin(), groups.end(), group_greater);
    enum { LEFT = 0, RIGHT = 1 };
    std::map<std::pair<int64_t, int>, int> used;      // presence + value; insertion order of the value-1 keys goes to keep_seq
    auto has = [&](int64_t k, int side) { return used.find({k, side}) != used.end(); };
    std::vector<Group> chosen;
    for (const Group &g : groups) {
        const int64_t first = g.ids.front(), last = g.ids.back();
        if (has(first, LEFT) || has(last, RIGHT)) continue;
        if (g.flag > 0 && (has(first, RIGHT) || has(last, LEFT))) continue;
        if (first != last) {
            const int64_t lo = std::min(first, last), hi = std::max(first, last);
            bool blocked = false;
            std::vector<int64_t> between;
            for (int64_t k = lo + 1; k < hi; ++k)
                if (m[k].r == m[first].r || m[k].r == m[last].r) {
                    between.push_back(k);
                    if (has(k, LEFT) || has(k, RIGHT)) blocked = true;
                }
            if (blocked) continue;
            for (int64_t k : between) { used[{k, LEFT}] = 0; used[{k, RIGHT}] = 0; }
        }
        chosen.push_back(g);
        used[{first, LEFT}] = 1; out.keep_seq.push_back(first);
        used[{last, RIGHT}] = 1; out.keep_seq.push_back(last);
        if (g.flag > 0) {
            used[{first, RIGHT}] = 1; out.keep_seq.push_back(first);
            used[{last, LEFT}] = 1; out.keep_seq.push_back(last);
        }
    }
    std::stable_sort(chosen.begin(), chosen.end(), [](const Group &a, const Group &b) { return a.ids.front() > b.ids.front(); });
    for (size_t k = 0; k + 1 < chosen.size(); ++k) {
        Group &g1 = chosen[k];
        Group &g2 = chosen[k + 1];
        if (g1.ids.front() == g2.ids.back()) {         // g2 ends with the hit g1 starts with: join the chains
            const Row &mm = m[g1.ids.front()];
            const int64_t span = mm.qe - mm.qs + 1;
            const int64_t length = g1.span + g2.span - span;
            const double iden = (g1.iden * (double)g1.span + g2.iden * (double)g2.span - std::min(g1.iden, g2.iden) * (double)span) / (double)length;
            Group joined{g1.score + g2.score - mm.score, iden, length, 0, {g2.ids.front()}};
            joined.ids.insert(joined.ids.end(), g1.ids.begin(), g1.ids.end());
            g1.iden = -1;
            chosen[k + 1] = std::move(joined);
        }
    }
    for (Group &g : chosen)
        if (g.iden >= 0) out.assign.push_back(std::move(g));
}

}  // namespace

extern "C" {

int pep_ovl_filter(uint64_t n, const int64_t *q, const int64_t *r, const int64_t *qs, const int64_t *qe, const int64_t *ss, const int64_t *se,
                   const double *score, double *iden, double coverage, double delta)
{
    if (n && (!q || !r || !qs || !qe || !ss || !se || !score || !iden)) return PEP_ERR_ARG;
    std::vector<uint64_t> losers;
    for (uint64_t i = 0; i < n; ++i) {
        if (iden[i] < 0) continue;
        const int64_t a_len = se[i] - ss[i] + 1, a_qlen = qe[i] - qs[i] + 1;
        losers.clear();
        for (uint64_t j = i + 1; j < n; ++j) {
            if (iden[j] < 0) continue;
            if (q[i] != q[j] || r[i] != r[j] || se[i] < ss[j]) break;
            const int64_t b_len = se[j] - ss[j] + 1, b_qlen = qe[j] - qs[j] + 1;
            const int64_t shared = std::min(se[i], se[j]) - ss[j] + 1;
            if ((double)shared >= coverage * (double)a_len && score[j] - score[i] >= delta) {
                iden[i] = -1.;
                break;
            } else if ((double)shared >= coverage * (double)b_len && score[i] - score[j] >= delta) {
                losers.push_back(j);
            } else if (shared >= a_len && (double)shared < coverage * (double)b_len) {
                const int64_t q_shared = std::min(qe[i], qe[j]) - std::max(qs[j], qs[i]) + 1;
                if (q_shared >= a_qlen && (double)q_shared < coverage * (double)b_qlen) break;   // the reference's no-op comparison (uberBlast.py:440)
            } else if (shared >= b_len && (double)shared < coverage * (double)a_len) {
                const int64_t q_shared = std::min(qe[i], qe[j]) - std::max(qs[j], qs[i]) + 1;
                if (q_shared >= b_qlen && (double)q_shared < coverage * (double)a_qlen) losers.push_back(j);
            }
        }
        if (iden[i] >= 0)
            for (uint64_t j : losers) iden[j] = -1.;
    }
    return PEP_OK;
}

int pep_linear_merge(uint64_t n, const int64_t *q, const int64_t *r, const double *iden, const int64_t *qs, const int64_t *qe, const int64_t *ss,
                     const int64_t *se, const double *score, const int64_t *ql, const int64_t *sl, const int64_t *rid, double gap_dist, double len_diff,
                     int64_t *keep_seq, uint64_t keep_cap, uint64_t *n_keep, uint64_t *query_off, uint8_t *query_ascending, uint64_t *n_query,
                     double *grp_score, double *grp_iden, int64_t *grp_span, uint64_t *grp_ids_off, int64_t *grp_ids, uint64_t ids_cap, uint64_t *n_ids)
{
    if (!n_keep || !n_query || !n_ids) return PEP_ERR_ARG;
    *n_keep = *n_query = *n_ids = 0;
    if (n == 0) return PEP_OK;
    if (!q || !r || !iden || !qs || !qe || !ss || !se || !score || !ql || !sl || !rid || !query_off || !query_ascending || !grp_score || !grp_iden ||
        !grp_span || !grp_ids_off)
        return PEP_ERR_ARG;
    std::vector<Row> rows(n);
    for (uint64_t i = 0; i < n; ++i) rows[i] = Row{q[i], r[i], qs[i], qe[i], ss[i], se[i], ql[i], sl[i], rid[i], iden[i], score[i]};
    // per row: the group it ends up in (later assignments win, like the reference's loop)
    std::vector<int64_t> row_group(n, -1);
    std::vector<Group> all;
    std::vector<uint64_t> all_base;
    std::vector<int64_t> seq;
    std::vector<uint64_t> q_off{0};
    std::vector<uint8_t> q_asc;
    for (uint64_t lo = 0; lo < n;) {
        uint64_t hi = lo + 1;
        while (hi < n && q[hi] == q[lo]) ++hi;
        MergeOut out;
        merge_one_query(rows.data() + lo, (int64_t)(hi - lo), gap_dist, len_diff, out);
        if (out.ascending)
            for (uint64_t k = lo; k < hi; ++k) seq.push_back((int64_t)k);
        else
            for (int64_t k : out.keep_seq) seq.push_back((int64_t)lo + k);
        q_asc.push_back(out.ascending ? 1 : 0);
        q_off.push_back(seq.size());
        for (Group &g : out.assign) {
            for (int64_t k : g.ids) row_group[lo + (uint64_t)k] = (int64_t)all.size();
            all.push_back(std::move(g));
            all_base.push_back(lo);
        }
        lo = hi;
    }
    uint64_t ids_total = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (row_group[i] >= 0) ids_total += all[(size_t)row_group[i]].ids.size();
    *n_keep = seq.size();
    *n_query = q_asc.size();
    *n_ids = ids_total;
    if (seq.size() > keep_cap || ids_total > ids_cap) return PEP_OK;       // caller re-calls with larger buffers
    if ((seq.size() && !keep_seq) || (ids_total && !grp_ids)) return PEP_ERR_ARG;
    std::copy(seq.begin(), seq.end(), keep_seq);
    std::copy(q_off.begin(), q_off.end(), query_off);
    std::copy(q_asc.begin(), q_asc.end(), query_ascending);
    uint64_t at = 0;
    for (uint64_t i = 0; i < n; ++i) {
        grp_ids_off[i] = at;
        if (row_group[i] < 0) { grp_score[i] = grp_iden[i] = 0; grp_span[i] = -1; continue; }
        const Group &g = all[(size_t)row_group[i]];
        grp_score[i] = g.score; grp_iden[i] = g.iden; grp_span[i] = g.span;
        for (int64_t k : g.ids) grp_ids[at++] = rid[all_base[(size_t)row_group[i]] + (uint64_t)k];
    }
    grp_ids_off[n] = at;
    return PEP_OK;
}

// compare_prediction over the columns of one genome's table (PEPPAN.py:869-901; mapbsn._with_known is its numpy statement and the test's yardstick): see peppan_hip.h
int pep_known_order(uint64_t n, const int64_t *ri, const int64_t *r_code, const int64_t *q_code, const int64_t *ss, const int64_t *se, const int64_t *qs,
                    const int64_t *qe, const int64_t *ql, const double *score, uint64_t n_contigs, const uint64_t *g_off, const int64_t *g1, const int64_t *g2,
                    const uint8_t *g_plus, const uint8_t *g_sorted, int64_t *order, double *known)
{
    if (n && (!ri || !r_code || !q_code || !ss || !se || !qs || !qe || !ql || !score || !g_off || !order || !known)) return PEP_ERR_ARG;
    auto mod3 = [](int64_t x) { const int64_t m = x % 3; return m < 0 ? m + 3 : m; };          // numpy's %: the sign of the divisor
    std::vector<int64_t> lo(n), hi(n), first(n);
    for (uint64_t i = 0; i < n; ++i) { lo[i] = std::min(ss[i], se[i]); hi[i] = std::max(ss[i], se[i]); first[i] = (int64_t)i; }
    // the order the reference walks the table in: (contig, lower reference coordinate), stable
    std::stable_sort(first.begin(), first.end(), [&](int64_t a, int64_t b) { return r_code[a] != r_code[b] ? r_code[a] < r_code[b] : lo[a] < lo[b]; });
    std::vector<double> kn(n, 0.1);                  // known[k] belongs to row first[k]
    std::vector<int64_t> cmax;
    for (uint64_t a = 0; a < n;) {
        uint64_t b = a + 1;
        while (b < n && ri[first[b]] == ri[first[a]]) ++b;
        const int64_t c = ri[first[a]];
        if (c < 0 || (uint64_t)c >= n_contigs) return PEP_ERR_ARG;
        const uint64_t g0 = g_off[c], ng = g_off[c + 1] - g_off[c];
        if (ng) {
            const int64_t *p1 = g1 + g0, *p2 = g2 + g0;
            const uint8_t *pl = g_plus + g0;
            auto judge = [&](uint64_t k, uint64_t j) {          // gene j against the row at position k of the walk
                const int64_t row = first[k], s = lo[row], e = hi[row];
                const bool fwd = ss[row] < se[row];
                const int64_t head = ss[row] - qs[row] + 1, tail = se[row] + (ql[row] - qe[row]);
                const int64_t f1 = fwd ? mod3(head) + 1 : mod3(-head) - 1, f2 = fwd ? mod3(tail + 1) + 1 : mod3(-(tail - 1)) - 1;
                const int64_t m1 = pl[j] ? mod3(p1[j]) + 1 : mod3(-(p1[j] - 1)) - 1, m2 = pl[j] ? mod3(p2[j] + 1) + 1 : mod3(-p2[j]) - 1;
                if (m1 != f1 && m1 != f2 && m2 != f1 && m2 != f2) return;
                const int64_t plen = p2[j] - p1[j] + 1;
                const double ovl = (double)(std::min(e, p2[j]) - std::max(s, p1[j]) + 1);
                if (ovl >= 0.6 * (double)plen || ovl >= 0.6 * (double)(e - s + 1)) kn[k] = std::max(kn[k], ovl / (double)plen);
            };
            if (g_sorted[c]) {
                // genes in start order: every gene from the first whose running maximum of ends reaches the hit's start to the last that starts at or before its end
                cmax.resize(ng);
                int64_t m = p2[0];
                for (uint64_t j = 0; j < ng; ++j) { m = std::max(m, p2[j]); cmax[j] = m; }
                for (uint64_t k = a; k < b; ++k) {
                    const int64_t row = first[k];
                    const uint64_t f = (uint64_t)(std::lower_bound(cmax.begin(), cmax.end(), lo[row]) - cmax.begin());
                    const uint64_t l = std::max<uint64_t>((uint64_t)(std::upper_bound(p1, p1 + ng, hi[row]) - p1), f);
                    for (uint64_t j = f; j < l; ++j) judge(k, j);
                }
            } else {
                // any order: the reference's pointer sweep, row by row
                uint64_t at = 0;
                for (uint64_t k = a; k < b; ++k) {
                    const int64_t row = first[k];
                    while (at < ng && lo[row] > p2[at]) ++at;
                    for (uint64_t j = at; j < ng; ++j) {
                        if (hi[row] < p1[j]) break;
                        judge(k, j);
                    }
                }
            }
        }
        a = b;
    }
    // the order compare_prediction returns the table in: (query, contig, score), stable on top of the walk
    std::vector<int64_t> again(n);
    for (uint64_t k = 0; k < n; ++k) again[k] = (int64_t)k;
    std::stable_sort(again.begin(), again.end(), [&](int64_t x, int64_t y) {
        const int64_t a = first[x], b = first[y];
        if (q_code[a] != q_code[b]) return q_code[a] < q_code[b];
        if (r_code[a] != r_code[b]) return r_code[a] < r_code[b];
        return score[a] < score[b];
    });
    for (uint64_t k = 0; k < n; ++k) { order[k] = first[again[k]]; known[k] = kn[again[k]]; }
    return PEP_OK;
}

}  // extern "C"
