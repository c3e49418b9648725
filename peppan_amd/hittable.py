"""Numeric form of the reference's "blastab" hit table (column layout: SURVEY.md section 8, uberBlast.py:57-58, 280-288, 354).

The reference keeps one Python object per cell from the moment a tool's output is parsed (uberBlast.py:67, 306) and every
post-processing step (reScore, -f, -m, fixEnd, -O, the final sort; uberBlast.py:352-376) walks rows of Python objects.  Here the
whole chain runs on flat numpy columns + one CIGAR arena (uint32 runs len << 2 | op, op 0 = M, 1 = I, 2 = D, nucleotide units) -
the layout the GPU hands over - and the 16/17-column object rows the callers expect are created ONCE, at the end of
RunBlast.run (`to_rows`).  `from_rows` is the way back in: tables that arrive as object rows (the reference's tool plug-in
contract, canned tables in the tests) take the same numeric chain.
"""
import ctypes as C
import os

import numpy as np

_PYROWS = None


def _pyrows():
    """the C pass that builds the object rows (csrc/pyrows.c -> _pyrows.so, through the CPython API with the GIL held)"""
    global _PYROWS
    if _PYROWS is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_pyrows.so')
        if not os.path.exists(path):
            raise RuntimeError('peppan_amd/_pyrows.so is not built (run `python -c "import __graft_entry__ as g; g.build()"` or `make -C peppan_amd/csrc`)')
        lib = C.PyDLL(path)
        lib.pep_rows_fill.restype = C.c_int
        lib.pep_rows_fill.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.py_object, C.py_object] + [C.c_void_p] * 12 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        lib.pep_genes_scan.restype = C.c_ssize_t
        lib.pep_genes_scan.argtypes = [C.py_object, C.py_object, C.py_object] + [C.c_void_p] * 5 + [C.c_ssize_t]
        lib.pep_strs_measure.restype = C.c_int64
        lib.pep_strs_measure.argtypes = [C.py_object, C.c_void_p]
        lib.pep_strs_pack.restype = C.c_int
        lib.pep_strs_pack.argtypes = [C.py_object, C.c_void_p, C.c_int64]
        lib.pep_digest_ints.restype = C.py_object
        lib.pep_digest_ints.argtypes = [C.c_void_p, C.c_ssize_t, C.c_ssize_t]
        lib.pep_records_dict.restype = C.py_object
        lib.pep_records_dict.argtypes = [C.c_char_p] + [C.c_void_p] * 4 + [C.c_ssize_t]
        _PYROWS = lib
    return _PYROWS


_OPS = np.array(['M', 'I', 'D'])
_OP_BYTES = np.frombuffer(b'MID', dtype=np.uint8)
_OP_CODE = {'M': 0, 'I': 1, 'D': 2}
_INT_TYPES = (int, np.integer)


_INT_TABLES = {}         # id(name table) -> (the table, its int64 array): the genomes of a mapping batch share one 10 000-entry table of gene ids


def int_name_array(names, register=None):
    """the int64 array of a name table of integers, remembered per table OBJECT (None when the table holds anything else).  register: the
    array of a table the caller has just converted itself"""
    hit = _INT_TABLES.get(id(names))
    if hit is not None and hit[0] is names:
        return hit[1]
    if register is None:
        if not (len(names) and all(isinstance(v, _INT_TYPES) for v in names)):
            return None
        register = np.asarray(names, dtype=np.int64)
    if len(names) >= 256:
        if len(_INT_TABLES) >= 16:
            _INT_TABLES.clear()
        _INT_TABLES[id(names)] = (names, register)
    return register


def _name_codes(names, idx):
    """integer code of every row's name in the order a sort of the column orders the names: numeric when every name is an integer,
    code-point order of the strings otherwise (what pandas / numpy do with an object column of that content)"""
    num = int_name_array(names)
    if num is not None:
        return num[idx]
    rank = np.unique(np.array([str(v) for v in names], dtype=str), return_inverse=True)[1].astype(np.int64) if len(names) else np.zeros(0, np.int64)
    return rank[idx]


class HitTable(object):
    __slots__ = ('q_tab', 'r_tab', 'qi', 'ri', 'iden', 'aln', 'mis', 'gap', 'qs', 'qe', 'ss', 'se', 'evalue', 'score', 'score_is_int',
                 'ql', 'sl', 'arena', 'c_off', 'c_runs', 'rid', 'merge', 'm_score', 'm_iden', 'm_span', 'm_start', 'm_len', 'm_ids', 'q_sorted', 'r_sorted', 'rescored')
    _ROW_COLS = ('qi', 'ri', 'iden', 'aln', 'mis', 'gap', 'qs', 'qe', 'ss', 'se', 'evalue', 'score', 'ql', 'sl', 'c_off', 'c_runs', 'rid')
    # column 16 after -m in numeric form: per row the group's score / identity / span (span < 0: the row has no group and shows the
    # reference's shared empty list) and the slice [m_start, m_start + m_len) of m_ids that holds the group's row ids
    _MERGE_COLS = ('m_score', 'm_iden', 'm_span', 'm_start', 'm_len')

    def __init__(self, q_tab, r_tab, qi, ri, iden, aln, mis, gap, qs, qe, ss, se, evalue, score, ql, sl, arena, c_off, c_runs,
                 rid=None, merge=None, score_is_int=True):
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.q_tab, self.r_tab = q_tab, r_tab
        self.qi, self.ri = i64(qi), i64(ri)
        self.iden, self.evalue, self.score = f64(iden), f64(evalue), f64(score)
        self.aln, self.mis, self.gap, self.qs, self.qe, self.ss, self.se, self.ql, self.sl = (i64(a) for a in (aln, mis, gap, qs, qe, ss, se, ql, sl))
        self.arena = np.ascontiguousarray(arena, dtype=np.uint32)
        self.c_off, self.c_runs = i64(c_off), i64(c_runs)
        self.rid = i64(rid) if rid is not None else np.full(len(self.qi), -1, dtype=np.int64)
        self.merge = merge                      # column 16 after -m as Python lists, one per row (made on demand from the m_* arrays)
        self.m_score = self.m_iden = self.m_span = self.m_start = self.m_len = self.m_ids = None
        self.score_is_int = score_is_int
        self.q_sorted = self.r_sorted = False   # True: the name table is a list of str in ascending code-point order (row codes = indices)
        self.rescored = False                   # True on a tool's table whose identity / score are reScore mode 1's already (the search counted the identical columns: Context.set_nt_match)

    def __len__(self):
        return len(self.qi)

    # ------------------------------------------------------------------------------------------------ construction
    @classmethod
    def empty(cls):
        z = np.zeros(0, dtype=np.int64)
        return cls([], [], z, z, z, z, z, z, z, z, z, z, z, z, z, z, np.zeros(0, np.uint32), z, z)

    @classmethod
    def from_rows(cls, rows):
        """object table [n, 15 | 16 | 17] (CIGAR as [[n, op], ...] lists or as a string) -> HitTable"""
        import re
        n = rows.shape[0]
        if n == 0:
            return cls.empty()
        q_tab, r_tab, q_of, r_of = [], [], {}, {}
        qi = np.fromiter((q_of.setdefault(v, len(q_of)) for v in rows[:, 0].tolist()), dtype=np.int64, count=n)
        ri = np.fromiter((r_of.setdefault(v, len(r_of)) for v in rows[:, 1].tolist()), dtype=np.int64, count=n)
        q_tab, r_tab = list(q_of), list(r_of)
        lens, ops, runs = [], [], np.zeros(n, dtype=np.int64)
        for k, c in enumerate(rows[:, 14].tolist()):
            if isinstance(c, str):
                c = [(int(a), b) for a, b in re.findall(r'(\d+)([MID])', c)]
            runs[k] = len(c)
            for a, b in c:
                lens.append(int(a))
                ops.append(_OP_CODE[b])
        arena = ((np.array(lens, dtype=np.int64) << 2) | np.array(ops, dtype=np.int64)).astype(np.uint32) if lens else np.zeros(0, np.uint32)
        off = np.concatenate([[0], np.cumsum(runs)[:-1]]) if n else runs
        num = lambda c, dt: np.ascontiguousarray(rows[:, c], dtype=dt)
        sc = rows[:, 11].tolist()
        t = cls(q_tab, r_tab, qi, ri, num(2, np.float64), num(3, np.int64), num(4, np.int64), num(5, np.int64), num(6, np.int64), num(7, np.int64),
                num(8, np.int64), num(9, np.int64), num(10, np.float64), num(11, np.float64), num(12, np.int64), num(13, np.int64), arena, off, runs,
                rid=num(15, np.int64) if rows.shape[1] > 15 else None,
                score_is_int=all(isinstance(v, _INT_TYPES) for v in sc))
        if rows.shape[1] > 16:
            t.set_merge_lists(rows[:, 16].tolist())
        return t

    def set_merge_lists(self, lists):
        """column 16 given as [score, identity, span, row ids...] lists (an empty list = no group)"""
        n = len(lists)
        self.merge = lists
        self.m_score = np.array([g[0] if len(g) else 0. for g in lists], dtype=np.float64)
        self.m_iden = np.array([g[1] if len(g) else 0. for g in lists], dtype=np.float64)
        self.m_span = np.array([g[2] if len(g) else -1 for g in lists], dtype=np.int64)
        self.m_len = np.array([max(0, len(g) - 3) for g in lists], dtype=np.int64)
        self.m_start = np.concatenate([[0], np.cumsum(self.m_len)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
        self.m_ids = np.array([i for g in lists for i in g[3:]], dtype=np.int64)

    def has_merge(self):
        return self.m_span is not None

    def merge_lists(self):
        if self.merge is None and self.m_span is not None:
            s_l, i_l, sp_l, ids = self.m_score.tolist(), self.m_iden.tolist(), self.m_span.tolist(), self.m_ids.tolist()
            shared = []
            self.merge = [[s_l[k], i_l[k], sp_l[k]] + ids[a:a + b] if sp_l[k] >= 0 else shared
                          for k, (a, b) in enumerate(zip(self.m_start.tolist(), self.m_len.tolist()))]
        return self.merge

    def take(self, idx):
        """rows idx (index array or boolean mask), in that order; name tables and the CIGAR arena are shared"""
        idx = np.asarray(idx)
        if idx.dtype == bool:
            idx = np.flatnonzero(idx)
        t = HitTable.__new__(HitTable)
        for f in self.__slots__:
            setattr(t, f, getattr(self, f))
        names = self._ROW_COLS + (self._MERGE_COLS if self.m_span is not None else ())
        from ._native import cols_gather
        for f, col in zip(names, cols_gather([np.ascontiguousarray(getattr(self, f)) for f in names], idx)):      # every column in one pass of host C++
            setattr(t, f, col)
        if self.merge is not None:
            t.merge = [self.merge[i] for i in idx.tolist()]
        return t

    @staticmethod
    def concat(tables):
        """rows of several tables one after the other (uberBlast.py:353 vstack); name tables and arenas are merged"""
        tables = [t for t in tables if len(t)]
        if not tables:
            return HitTable.empty()
        if len(tables) == 1:
            return tables[0]
        if all(t.q_tab is tables[0].q_tab and t.r_tab is tables[0].r_tab for t in tables):
            # the tools of one run share their name tables: columns one after the other, arenas side by side
            base = np.cumsum([0] + [len(t.arena) for t in tables[:-1]])
            cols = {f: np.concatenate([getattr(t, f) for t in tables]) for f in HitTable._ROW_COLS if f != 'c_off'}
            first = tables[0]
            out = HitTable(first.q_tab, first.r_tab, cols['qi'], cols['ri'], cols['iden'], cols['aln'], cols['mis'], cols['gap'], cols['qs'], cols['qe'], cols['ss'], cols['se'],
                           cols['evalue'], cols['score'], cols['ql'], cols['sl'], np.concatenate([t.arena for t in tables]),
                           np.concatenate([t.c_off + b for t, b in zip(tables, base.tolist())]), cols['c_runs'], rid=cols['rid'],
                           score_is_int=all(t.score_is_int for t in tables))
            out.q_sorted, out.r_sorted = all(t.q_sorted for t in tables), all(t.r_sorted for t in tables)
            return out
        q_of, r_of, parts = {}, {}, {f: [] for f in HitTable._ROW_COLS}
        arena, base = [], 0
        for t in tables:
            qmap = np.fromiter((q_of.setdefault(v, len(q_of)) for v in t.q_tab), dtype=np.int64, count=len(t.q_tab))
            rmap = np.fromiter((r_of.setdefault(v, len(r_of)) for v in t.r_tab), dtype=np.int64, count=len(t.r_tab))
            for f in HitTable._ROW_COLS:
                parts[f].append(getattr(t, f))
            parts['qi'][-1] = qmap[t.qi]
            parts['ri'][-1] = rmap[t.ri]
            parts['c_off'][-1] = t.c_off + base
            arena.append(t.arena)
            base += len(t.arena)
        cols = {f: np.concatenate(v) for f, v in parts.items()}
        out = HitTable(list(q_of), list(r_of), cols['qi'], cols['ri'], cols['iden'], cols['aln'], cols['mis'], cols['gap'], cols['qs'], cols['qe'],
                       cols['ss'], cols['se'], cols['evalue'], cols['score'], cols['ql'], cols['sl'], np.concatenate(arena), cols['c_off'], cols['c_runs'],
                       rid=cols['rid'], score_is_int=all(t.score_is_int for t in tables))
        return out

    # ------------------------------------------------------------------------------------------------ pieces of the chain
    def q_codes(self):
        return self.qi if self.q_sorted and type(self.q_tab[0]) is str else _name_codes(self.q_tab, self.qi)

    def r_codes(self):
        return self.ri if self.r_sorted and type(self.r_tab[0]) is str else _name_codes(self.r_tab, self.ri)

    def fix_end(self, se_lim, ee_lim):
        """RunBlast.fixEnd (uberBlast.py:462-480): stretch an alignment over an unaligned query head of at most se_lim bases / tail of at
        most ee_lim bases, as far as the reference sequence allows; the first / last CIGAR run grows by the same amount whatever its
        operation is; identity, score and columns 3-5 stay.  All rows at once."""
        if not len(self):
            return
        from ._native import cols_fix_end
        cols = {f: np.array(getattr(self, f)) for f in ('qs', 'qe', 'ss', 'se', 'c_off')}          # (the columns that change: private copies, rows may be shared with the caller's table)
        cols.update({f: getattr(self, f) for f in ('ql', 'sl', 'c_runs')})
        self.arena = cols_fix_end(cols, self.arena, se_lim, ee_lim)                                  # host C++; the rows' runs in a private arena
        self.qs, self.qe, self.ss, self.se, self.c_off = (cols[f] for f in ('qs', 'qe', 'ss', 'se', 'c_off'))

    def final_order(self):
        """the sort that ends RunBlast.run (uberBlast.py:375): by query name, reference name (as the column's values sort: strings
        lexicographically - '10' < '9'), then score; stable"""
        q, r = self.q_codes(), self.r_codes()
        if len(q) and min(int(q.min()), int(r.min())) >= 0:
            from ._native import cols_order
            return cols_order(q, r, self.score)              # host C++: radix passes over the two codes, the rows of one pair by score
        return np.lexsort((self.score, r, q))

    # ------------------------------------------------------------------------------------------------ object rows
    def cigar_strings(self):
        """'150M3D150M' per row (uberBlast.py:480), built for all rows at once"""
        n = len(self)
        if n == 0:
            return []
        runs = self.c_runs
        start = np.concatenate([[0], np.cumsum(runs)])
        src = np.repeat(self.c_off - start[:-1], runs) + np.arange(int(start[-1]))
        a = self.arena[src].astype(np.int64)
        lens, ops = a >> 2, a & 3
        nd = np.ones(len(a), dtype=np.int64)
        p = 10
        while len(lens) and (lens >= p).any():
            nd += lens >= p
            p *= 10
        width = nd + 1
        pos = np.concatenate([[0], np.cumsum(width)])
        buf = np.empty(int(pos[-1]), dtype=np.uint8)
        k, p = 0, 1
        while len(lens) and (nd > k).any():
            m = nd > k
            buf[pos[:-1][m] + nd[m] - 1 - k] = 48 + (lens[m] // p) % 10
            k += 1
            p *= 10
        buf[pos[:-1] + nd] = _OP_BYTES[ops]
        text = buf.tobytes().decode('ascii')
        row_pos = pos[start].tolist()
        return [text[row_pos[i]:row_pos[i + 1]] for i in range(n)]

    def cigar_lists(self):
        """[[n, op], ...] per row - the form the tools hand over in the reference (uberBlast.py:33, 316-319)"""
        a = self.arena.astype(np.int64)
        pairs = list(map(list, zip((a >> 2).tolist(), _OPS[a & 3].tolist())))
        return [pairs[o:o + r] for o, r in zip(self.c_off.tolist(), self.c_runs.tolist())]

    def to_rows(self, cigar='list', with_rid=True):
        """ndarray(object)[n, 15 | 16 | 17]: Python scalars per cell, CIGAR as lists ('list') or as text ('str').  One C pass over the
        columns (csrc/pyrows.c); only column 16 after -m (lists of Python objects per merged group) is filled from here."""
        n = len(self)
        width = 15 + (1 if with_rid else 0) + (1 if (self.has_merge() and with_rid) else 0)
        out = np.empty([n, width], dtype=object)
        if n == 0:
            return out
        q_tab, r_tab = (self.q_tab if isinstance(self.q_tab, list) else list(self.q_tab)), (self.r_tab if isinstance(self.r_tab, list) else list(self.r_tab))
        cols = [np.ascontiguousarray(a, dtype=dt) for a, dt in ((self.qi, np.int64), (self.ri, np.int64), (self.iden, np.float64), (self.aln, np.int64),
                (self.mis, np.int64), (self.gap, np.int64), (self.qs, np.int64), (self.qe, np.int64), (self.ss, np.int64), (self.se, np.int64),
                (self.evalue, np.float64), (self.score, np.float64))]
        tail = [np.ascontiguousarray(a, dtype=dt) for a, dt in ((self.ql, np.int64), (self.sl, np.int64), (self.arena if len(self.arena) else np.zeros(1, np.uint32), np.uint32),
                (self.c_off, np.int64), (self.c_runs, np.int64))]
        rid = np.ascontiguousarray(self.rid, dtype=np.int64) if with_rid else None
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = _pyrows().pep_rows_fill(C.c_void_p(out.ctypes.data), n, width, q_tab, r_tab, *[ptr(a) for a in cols], 1 if self.score_is_int else 0,
                                     *[ptr(a) for a in tail], 1 if cigar == 'str' else 2, ptr(rid) if rid is not None else None)
        if rc != 0:
            raise RuntimeError('pep_rows_fill failed')
        if with_rid and self.has_merge():
            col = out[:, 16]
            for k, v in enumerate(self.merge_lists()):
                col[k] = v
        return out
