"""Post-filters of the genes->genome mapping calls (flags -f, -m, -O of uberBlast; PEPPAN.py:768-772).

Host-side restatements of RunBlast.ovlFilter (uberBlast.py:417-452), RunBlast.linearMerge + _linearMerge
(uberBlast.py:100-218, 453-460) and RunBlast.returnOverlap + tab2overlaps (uberBlast.py:73-97, 378-395), pinned
by tests/golden/g07_filters.json.  They work on the 16-column object table (SURVEY.md section 8 layout):
  0 query, 1 reference, 2 identity, 3 aln length, 4 mismatches, 5 gap opens, 6/7 query start/end,
  8/9 reference start/end (start > end = reverse strand), 10 e-value, 11 score, 12/13 query/reference length,
  14 CIGAR, 15 row id, 16 (after merging) [score, identity, span, row ids...]
"""
from operator import itemgetter

import numpy as np
import pandas as pd

Q, R, IDEN, QS, QE, SS, SE, SCORE, QLEN, SLEN, RID = 0, 1, 2, 6, 7, 8, 9, 11, 12, 13, 15


def _fold_strand(tab):
    """reverse-strand hits get negative reference coordinates so that start < end everywhere"""
    rev = tab.T[SS] > tab.T[SE]
    tab[rev, SS:SE + 1] *= -1


def _unfold_strand(tab):
    neg = tab.T[SS] < 0
    tab[neg, SS:SE + 1] *= -1


# ------------------------------------------------------------------------------------------------ -f
def ovl_filter_py(blastab, coverage, delta):
    """pure-Python statement of ovl_filter (kept as the cross-check of the C++ port in the tests)
    drop the weaker of two hits of the same (query, reference) that overlap >= `coverage` on the reference"""
    _fold_strand(blastab)
    tab = pd.DataFrame(blastab).sort_values(by=[R, Q, SS, QS]).values
    n = tab.shape[0]
    for i in range(n):
        a = tab[i]
        if a[IDEN] < 0:
            continue
        a_len, a_qlen = a[SE] - a[SS] + 1, a[QE] - a[QS] + 1
        losers = []
        for j in range(i + 1, n):
            b = tab[j]
            if b[IDEN] < 0:
                continue
            if a[Q] != b[Q] or a[R] != b[R] or a[SE] < b[SS]:
                break
            b_len, b_qlen = b[SE] - b[SS] + 1, b[QE] - b[QS] + 1
            shared = min(a[SE], b[SE]) - b[SS] + 1
            if shared >= coverage * a_len and b[SCORE] - a[SCORE] >= delta:
                a[IDEN] = -1.
                break
            elif shared >= coverage * b_len and a[SCORE] - b[SCORE] >= delta:
                losers.append(j)
            elif shared >= a_len and shared < coverage * b_len:
                q_shared = min(a[QE], b[QE]) - max(b[QS], a[QS]) + 1
                if q_shared >= a_qlen and q_shared < coverage * b_qlen:
                    break          # the reference compares instead of assigning here (uberBlast.py:440): `a` survives
            elif shared >= b_len and shared < coverage * a_len:
                q_shared = min(a[QE], b[QE]) - max(b[QS], a[QS]) + 1
                if q_shared >= b_qlen and q_shared < coverage * a_qlen:
                    losers.append(j)
        if a[IDEN] >= 0:
            for j in losers:
                tab[j][IDEN] = -1.
    tab = tab[tab.T[IDEN] >= 0]
    _unfold_strand(tab)
    return tab


# ------------------------------------------------------------------------------------------------ -m
def _pair_score(m1, m2, span1, span2, overlap):
    """score / identity of two chained hits; `overlap` = sorted (desc) pair of overlaps, negative = gap"""
    if overlap[0] > 0:
        score = m1[SCORE] + m2[SCORE] - overlap[0] * min(float(m1[SCORE]) / span1, float(m2[SCORE]) / span2)
        ident = (m1[IDEN] * span1 + m2[IDEN] * span2 - overlap[0] * min(m1[IDEN], m2[IDEN])) / (span1 + span2 - overlap[0])
    else:
        score = m1[SCORE] + m2[SCORE]
        ident = (m1[IDEN] * span1 + m2[IDEN] * span2) / (span1 + span2)
    if overlap[1] < 0:
        score += overlap[1] / 3.
    return score, ident


def _merge_one_query(matches, gap_dist, len_diff):
    """all hits of ONE query gene, sorted by (reference, folded start, query start): chain collinear neighbours"""
    extra = np.empty((matches.shape[0], 1), dtype=object)
    extra.fill([])                                  # one shared empty list, as pd.Series([[]] * n) gives the reference
    matches = np.hstack([matches, extra])
    tail = 20
    n = len(matches)
    groups = []
    head_edge, tail_edge = [], []          # hits that run into a contig end (fragmented genes across contigs)
    for i, m1 in enumerate(matches):
        span1 = m1[QE] - m1[QS] + 1
        groups.append([m1[SCORE], m1[IDEN], span1, 0, i])
        if m1[QS] > tail and ((m1[SS] > 0 and m1[SS] - 1 <= gap_dist) or (m1[SS] < 0 and m1[SLEN] + m1[SS] < gap_dist)):
            tail_edge.append([i, m1])
        if m1[QE] <= m1[QLEN] - tail:
            if (m1[SS] > 0 and m1[SLEN] - m1[SE] <= gap_dist) or (m1[SS] < 0 and -1 - m1[SE] < gap_dist):
                head_edge.append([i, m1])
            for j in range(i + 1, n):
                m2 = matches[j]
                if m1[R] != m2[R] or (m1[SS] < 0 and m2[SS] > 0) or m2[SS] - m1[SE] - 1 >= gap_dist:
                    break
                q_span, r_span = m2[QE] - m1[QS] + 1, m2[SE] - m1[SS] + 1
                if abs(m1[IDEN] - m2[IDEN]) > 0.3 or m1[SS] + 3 >= m2[SS] or m1[SE] + 3 >= m2[SE] or m1[QS] + 3 >= m2[QS] \
                        or m1[QE] + 3 >= m2[QE] or m2[QS] - m1[QE] - 1 >= gap_dist or min(q_span, r_span) * len_diff < max(q_span, r_span):
                    continue
                span2 = m2[QE] - m2[QS] + 1
                overlap = sorted([m1[QE] - m2[QS] + 1, m1[SE] - m2[SS] + 1], reverse=True)
                score, ident = _pair_score(m1, m2, span1, span2, overlap)
                if score > m1[SCORE] and score > m2[SCORE]:
                    groups.append([score, ident, q_span, 0, i, j])
    if head_edge and tail_edge:
        for i, m1 in head_edge:
            for j, m2 in tail_edge:
                if (m1[R] == m2[R] and max(abs(m1[SS]), abs(m1[SE])) > min(abs(m2[SS]), abs(m2[SE]))) or abs(m1[IDEN] - m2[IDEN]) > 0.3 \
                        or m1[QS] >= m2[QS] or m1[QE] >= m2[QE] or m2[QS] - m1[QE] - 1 >= gap_dist:
                    continue
                q_span = m2[QE] - m1[QS] + 1
                g1 = -m1[SE] - 1 if m1[SE] < 0 else m1[SLEN] - m1[SE]
                g2 = m2[SS] - 1 if m2[SS] > 0 else m2[SLEN] + m2[SS]
                r_span = m1[SE] - m1[SS] + 1 + m2[SE] - m2[SS] + 1 + g1 + g2
                if g1 + g2 >= gap_dist or min(q_span, r_span) * len_diff < max(q_span, r_span):
                    continue
                overlap = sorted([m1[QE] - m2[QS] + 1, -g1 - g2], reverse=True)
                score, ident = _pair_score(m1, m2, m1[QE] - m1[QS] + 1, m2[QE] - m2[QS] + 1, overlap)
                if score > m1[SCORE] and score > m2[SCORE]:
                    groups.append([score, ident, q_span, 1, i, j])
    if len(groups) > n:
        groups.sort(reverse=True)
        used, chosen = {}, []
        LEFT, RIGHT = 4, 5                 # a hit can be used once as the left and once as the right part of a chain
        for g in groups:
            first, last = g[4], g[-1]
            if (first, LEFT) in used or (last, RIGHT) in used:
                continue
            if g[3] > 0 and ((first, RIGHT) in used or (last, LEFT) in used):
                continue
            if first != last:
                lo, hi = sorted([first, last])
                refs = {matches[first][R], matches[last][R]}
                between = [k for k in range(lo + 1, hi) if matches[k][R] in refs]
                if any((k, LEFT) in used or (k, RIGHT) in used for k in between):
                    continue
                for k in between:
                    used[(k, LEFT)] = used[(k, RIGHT)] = 0
            chosen.append(g)
            used[(first, LEFT)] = used[(last, RIGHT)] = 1
            if g[3] > 0:
                used[(first, RIGHT)] = used[(last, LEFT)] = 1
        chosen.sort(key=itemgetter(4), reverse=True)
        for k in range(len(chosen) - 1):
            g1, g2 = chosen[k:k + 2]
            if g1[4] == g2[-1]:             # ... g2 ends with the hit that g1 starts with: join the chains
                m = matches[g1[4]]
                span = m[QE] - m[QS] + 1
                length = g1[2] + g2[2] - span
                iden = (g1[1] * g1[2] + g2[1] * g2[2] - min(g1[1], g2[1]) * span) / length
                chosen[k + 1] = [g1[0] + g2[0] - m[SCORE], iden, length, 0, g2[4]] + g1[4:]
                g1[1] = -1
    else:
        chosen = groups
        used = {(k, k): 1 for k in np.arange(n)}
    for g in chosen:
        if g[1] >= 0:
            ids = [matches[k][RID] for k in g[4:]]
            for k in g[4:]:
                matches[k, -1] = g[:3] + ids
    keep = {k[0] for k, v in used.items() if v == 1}
    return matches[np.array(list(keep))]


def linear_merge_py(blastab, gap_dist, len_diff):
    """pure-Python statement of linear_merge (kept as the cross-check of the C++ port in the tests)"""
    _fold_strand(blastab)
    tab = pd.DataFrame(blastab).sort_values([Q, R, SS, QS]).values
    qid = np.unique(tab.T[Q], return_inverse=True)[1]
    parts = np.split(tab, np.where(np.diff(qid))[0] + 1)
    tab = np.vstack([_merge_one_query(p, gap_dist, len_diff) for p in parts])
    _unfold_strand(tab)
    return tab


# ------------------------------------------------------------------------------------------------ C++ ports (libpeppan_hip.so)
# The product path: numeric columns of a HitTable in, HitTable out; the object-table functions below wrap them for callers (and
# tests) that hold the reference's row format.
def _folded(T):
    """reference coordinates with reverse-strand hits negated, so that start < end everywhere (uberBlast.py:420, 455)"""
    rev = T.ss > T.se
    return np.where(rev, -T.ss, T.ss), np.where(rev, -T.se, T.se)


def ovl_filter_table(T, coverage, delta):
    """flag -f (RunBlast.ovlFilter, uberBlast.py:417-452): drop the weaker of two hits of the same (query, reference) that overlap
    >= `coverage` on the reference.  Rows come back in the (reference, query, start, query start) order the filter walks them in.
    The greedy pass itself is pep_ovl_filter (host C++ in libpeppan_hip.so)."""
    from . import _native as N
    if len(T) == 0:
        return T
    ss, se = _folded(T)
    q, r = T.q_codes(), T.r_codes()
    order = np.lexsort((T.qs, ss, q, r))
    iden = T.iden[order].copy()
    N.ovl_filter(q[order], r[order], T.qs[order], T.qe[order], ss[order], se[order], T.score[order], iden, coverage, delta)
    return T.take(order[iden >= 0])


def linear_merge_table(T, gap_dist, len_diff):
    """flag -m (RunBlast.linearMerge + _linearMerge, uberBlast.py:100-218, 453-460): chain collinear hits of one gene; column 16 =
    [score, identity, span, row ids...].  The chaining is pep_linear_merge (host C++); the row order inside a query that chained
    something is the iteration order of a Python set, as in the reference, so that part stays a set comprehension here."""
    from . import _native as N
    n = len(T)
    if n == 0:
        T.set_merge_lists([])
        return T
    ss, se = _folded(T)
    q, r = T.q_codes(), T.r_codes()
    first = np.lexsort((T.qs, ss, r, q))
    q = q[first]
    keep, q_off, asc, g_score, g_iden, g_span, ids_off, ids = N.linear_merge(q, r[first], T.iden[first], T.qs[first], T.qe[first], ss[first], se[first],
                                                                              T.score[first], T.ql[first], T.sl[first], T.rid[first], gap_dist, len_diff)
    if asc.all():
        order = keep
    else:
        first_row = np.concatenate([[0], np.flatnonzero(np.diff(q)) + 1])
        parts = []
        for k in range(len(asc)):
            seq = keep[q_off[k]:q_off[k + 1]]
            if asc[k]:
                parts.append(seq)
            else:
                lo = first_row[k]
                local = (seq - lo).tolist()
                parts.append(np.array(list({i for i in local}), dtype=np.int64) + lo)
        order = np.concatenate(parts)
    out = T.take(first[order])
    out.merge = None
    out.m_score, out.m_iden, out.m_span = g_score[order], g_iden[order], g_span[order]
    out.m_start, out.m_len, out.m_ids = ids_off[:-1][order], np.diff(ids_off)[order], np.asarray(ids, dtype=np.int64)
    return out


def overlaps_table(T, ovl_l, ovl_p, batch=1000000, sweep=None):
    """flag -O (RunBlast.returnOverlap + tab2overlaps, uberBlast.py:73-97, 378-395): pairs of hits (row ids) whose reference intervals
    overlap by >= min(ovl_l, ovl_p*len1) or >= ovl_p*len2.  The reference sweeps in batches of 1e6 pairs and, on resuming, emits
    the last pair of a full batch again (uberBlast.py:76-92); reproduced.  `sweep(contig, start, end, row_id, ovl_l, ovl_p)` runs
    the sweep itself (the GPU kernel K11 in the product); without it a plain Python loop does (used by the CPU tests)."""
    # intervals [contig id, row id, start, end] sorted by (contig, start, end), ties in table order; the reference numbers a contig
    # by the LAST row that names it (dict comprehension, uberBlast.py:380)
    n = len(T)
    if n:
        last = np.zeros(int(T.ri.max()) + 1, dtype=np.int64)
        np.maximum.at(last, T.ri, np.arange(n))
        lo, hi, cid = np.minimum(T.ss, T.se), np.maximum(T.ss, T.se), last[T.ri]
        iv = np.stack([cid, T.rid, lo, hi], axis=1)[np.lexsort((hi, lo, cid))].astype(int)
    else:
        iv = np.empty((0, 4), dtype=int)
    if sweep is not None and len(iv):
        res = np.asarray(sweep(iv[:, 0], iv[:, 2], iv[:, 3], iv[:, 1], float(ovl_l), float(ovl_p)), dtype=int).reshape(-1, 3)
        if len(res) >= batch:
            # the reference's resume quirk: the pair that fills a batch is emitted again as the first of the next one, so every later
            # batch holds batch - 1 new pairs: the duplicated pairs are number batch-1, 2*(batch-1), 3*(batch-1), ... (0-based)
            dup = np.arange(batch - 1, len(res), batch - 1)
            res = np.insert(res, dup + 1, res[dup], axis=0)
        return res[res.T[2] > 0]
    out = []
    n = len(iv)
    emitted = 0
    for i in range(n):
        c1, id1, s1, e1 = iv[i]
        need = min(ovl_l, ovl_p * (e1 - s1 + 1))
        for j in range(i + 1, n):
            c2, id2, s2, e2 = iv[j]
            if c1 != c2 or s2 > e1:
                break
            ovl = min(e1, e2) - s2 + 1
            if ovl >= need or ovl >= ovl_p * (e2 - s2 + 1):
                out.append([id1, id2, ovl])
                emitted += 1
                if emitted % batch == 0:            # the batch is full: the resumed sweep emits this pair once more, and it counts
                    out.append([id1, id2, ovl])     # towards the next batch
                    emitted += 1
    res = np.array(out, dtype=int).reshape(-1, 3)
    return res[res.T[2] > 0]


def _as_table(blastab):
    from .hittable import HitTable
    return blastab if isinstance(blastab, HitTable) else HitTable.from_rows(blastab)


def ovl_filter(blastab, coverage, delta):
    """object-table front of ovl_filter_table (same row format in and out)"""
    if blastab.shape[0] == 0:
        return blastab
    return ovl_filter_table(_as_table(blastab), coverage, delta).to_rows()


def linear_merge(blastab, gap_dist, len_diff):
    """object-table front of linear_merge_table: returns the rows with column 16 appended"""
    if blastab.shape[0] == 0:
        return np.hstack([blastab, np.empty((0, 1), dtype=object)])
    return linear_merge_table(_as_table(blastab), gap_dist, len_diff).to_rows()


def overlaps(blastab, ovl_l, ovl_p, batch=1000000, sweep=None):
    """object-table front of overlaps_table"""
    return overlaps_table(_as_table(blastab), ovl_l, ovl_p, batch, sweep)
