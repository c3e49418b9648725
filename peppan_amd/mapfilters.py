"""Post-filters of the genes->genome mapping calls (flags -f, -m, -O of uberBlast; PEPPAN.py:768-772).

Fronts of the library's RunBlast.ovlFilter (uberBlast.py:417-452), RunBlast.linearMerge + _linearMerge (uberBlast.py:100-218,
453-460) and RunBlast.returnOverlap + tab2overlaps (uberBlast.py:73-97, 378-395) replacements: the greedy passes are host C++
(csrc/mapfilters.hip), the interval sweep is K11 on the GPU.  Pinned by tests/golden/g07_filters.json (crafted cases) and
g18_filters_random.json.gz (200 random tables through the reference's own functions).  Column layout of the 16-column object table
(SURVEY.md section 8):
  0 query, 1 reference, 2 identity, 3 aln length, 4 mismatches, 5 gap opens, 6/7 query start/end,
  8/9 reference start/end (start > end = reverse strand), 10 e-value, 11 score, 12/13 query/reference length,
  14 CIGAR, 15 row id, 16 (after merging) [score, identity, span, row ids...]
"""
import numpy as np


# The product path: numeric columns of a HitTable in, HitTable out; the object-table functions at the end wrap them for callers (and
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
    order = N.lex_order((T.qs, ss, q, r))                    # (= np.lexsort: radix passes in host C++)
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
    first = N.lex_order((T.qs, ss, r, q))                    # (= np.lexsort)
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


def _intervals(T):
    """intervals [contig id, row id, start, end] sorted by (contig, start, end), ties in table order; the reference numbers a contig
    by the LAST row that names it (dict comprehension, uberBlast.py:380)"""
    n = len(T)
    if n == 0:
        return np.empty((0, 4), dtype=int)
    last = np.zeros(int(T.ri.max()) + 1, dtype=np.int64)
    np.maximum.at(last, T.ri, np.arange(n))
    lo, hi, cid = np.minimum(T.ss, T.se), np.maximum(T.ss, T.se), last[T.ri]
    from . import _native as N
    return np.stack([cid, T.rid, lo, hi], axis=1)[N.lex_order((hi, lo, cid))].astype(int)


def _swept(res, batch):
    """the pairs of one table's sweep as the reference returns them"""
    if len(res) >= batch:
        # the reference's resume quirk: the pair that fills a batch is emitted again as the first of the next one, so every later
        # batch holds batch - 1 new pairs: the duplicated pairs are number batch-1, 2*(batch-1), 3*(batch-1), ... (0-based)
        dup = np.arange(batch - 1, len(res), batch - 1)
        res = np.insert(res, dup + 1, res[dup], axis=0)
    return res[res.T[2] > 0]


def overlaps_tables(tables, ovl_l, ovl_p, sweep, batch=1000000):
    """overlaps_table for several tables (the genomes of one batched search) with ONE sweep: the tables' intervals behind one another, contig numbers and
    row ids shifted table by table - pairs never cross contigs, so they come out table by table and are cut and shifted back.  A sweep is two launches and two
    round trips; on a GPU that eight worker processes share each of them waits for its turn."""
    ivs = [_intervals(T) for T in tables]
    # contig numbers are row indices of their table (below its length), row ids are whatever the table carries (below its largest + 1)
    c_base = np.concatenate([[0], np.cumsum([len(T) for T in tables])]).astype(np.int64)
    r_base = np.concatenate([[0], np.cumsum([int(T.rid.max()) + 1 if len(T) else 0 for T in tables])]).astype(np.int64)
    big = [iv + np.array([c, r, 0, 0]) for iv, c, r in zip(ivs, c_base[:-1].tolist(), r_base[:-1].tolist()) if len(iv)]
    if not big:
        return [np.zeros((0, 3), dtype=int) for _ in tables]
    iv = np.concatenate(big)
    res = np.asarray(sweep(iv[:, 0], iv[:, 2], iv[:, 3], iv[:, 1], float(ovl_l), float(ovl_p)), dtype=int).reshape(-1, 3)
    owner = np.searchsorted(r_base, res[:, 0], side='right') - 1          # the table of every pair: the sweep goes contig by contig, so table by table
    in_order = len(res) == 0 or bool((np.diff(owner) >= 0).all())
    cuts = np.searchsorted(owner, np.arange(len(tables) + 1)) if in_order else None
    out = []
    for k in range(len(tables)):
        part = res[cuts[k]:cuts[k + 1]] if in_order else res[owner == k]
        out.append(_swept(part - np.array([r_base[k], r_base[k], 0]), batch))
    return out


def overlaps_table(T, ovl_l, ovl_p, batch=1000000, sweep=None):
    """flag -O (RunBlast.returnOverlap + tab2overlaps, uberBlast.py:73-97, 378-395): pairs of hits (row ids) whose reference intervals
    overlap by >= min(ovl_l, ovl_p*len1) or >= ovl_p*len2.  The reference sweeps in batches of 1e6 pairs and, on resuming, emits
    the last pair of a full batch again (uberBlast.py:76-92); reproduced.  `sweep(contig, start, end, row_id, ovl_l, ovl_p)` runs
    the sweep itself (the GPU kernel K11 in the product); without it a plain Python loop does (used by the CPU tests)."""
    iv = _intervals(T)
    if sweep is not None and len(iv):
        return _swept(np.asarray(sweep(iv[:, 0], iv[:, 2], iv[:, 3], iv[:, 1], float(ovl_l), float(ovl_p)), dtype=int).reshape(-1, 3), batch)
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
