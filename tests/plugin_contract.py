"""Test helper for SURVEY.md section 8(b): what a caller that owns the `tools` dictionary may rely on.

A tool of RunBlast is `method(ref, qry) -> ndarray(object)[n, 15]`: names as str, numbers as Python / numpy scalars, the CIGAR of
column 14 as a list of [length, 'M' | 'I' | 'D'] pairs, rows in any order, n = 0 allowed.  Tables of several tools can be stacked, take a
sixteenth column of row ids and go through the object-row forms of the post-processing methods (reScore, fixEnd, returnOverlap).  The
checks are stated here directly; that the whole amounts to what RunBlast.run returns is asserted by the tests that use them (and the
run() of the reference itself is held by golden G8)."""
import numpy as np


def assert_tool_table(t):
    """one tool's return value against the contract"""
    assert isinstance(t, np.ndarray) and t.dtype == object and t.ndim == 2 and t.shape[1] == 15, (type(t), getattr(t, 'shape', None))
    for row in t[:64]:
        assert type(row[0]) is str and type(row[1]) is str
        assert 0. <= float(row[2]) <= 1. and all(int(row[c]) == row[c] for c in (3, 4, 5, 6, 7, 8, 9, 12, 13))
        runs = row[14]
        assert type(runs) is list and runs and all(type(r[0]) is int and r[0] > 0 and r[1] in ('M', 'I', 'D') for r in runs)
        assert sum(r[0] for r in runs) == row[3]                                          # the alignment length is the CIGAR's


def through_public_methods(rb, tables, ref, qry, re_score=0, fix_end=(6., 6.), overlap=None):
    """tool tables -> one table with ids -> the public object-row methods -> ordered by (query, reference, score).  Returns the table
    (and the overlaps when `overlap` = (length, proportion) is given)."""
    filled = [t for t in tables if len(t)]
    if not filled:
        return np.empty([0, 16], dtype=object)
    n = sum(len(t) for t in filled)
    table = np.empty([n, 16], dtype=object)
    table[:, :15] = np.concatenate(filled, axis=0)
    table[:, 15] = list(range(n))
    if re_score:
        table = rb.reScore(ref, qry, table, re_score, rb.min_id, rb.table_id)
    rb.fixEnd(table, fix_end[0], fix_end[1])
    order = sorted(range(len(table)), key=lambda i: (table[i, 0], table[i, 1], table[i, 11]))     # (stable, like the multi-column sort of run())
    if overlap is not None:
        return table[order], rb.returnOverlap(table, [True, overlap[0], overlap[1]])
    return table[order]


def tools_then_methods(rb, ref, qry, methods, min_id, min_cov, min_ratio, table_id=11, **post):
    """the public tools called one by one, each held to the contract, then through_public_methods"""
    rb.min_id, rb.min_cov, rb.min_ratio, rb.table_id = min_id, min_cov, min_ratio, table_id
    by_name = {'blastn': rb.runBlast, 'diamond': rb.runDiamond, 'diamondself': rb.runDiamondSELF}
    tables = [by_name[m.lower()](ref, qry) for m in methods]
    for t in tables:
        assert_tool_table(t)
    return through_public_methods(rb, tables, ref, qry, **post)
