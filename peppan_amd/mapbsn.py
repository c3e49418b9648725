"""The caller and the on-disk formats on the far side of the genes->genomes mapping search.

    MapBsn              PEPPAN.py:27-114    zip archive whose members are .npy payloads keyed by str(key)
    decodeSeq           PEPPAN.py:318-324   base-5 triple packing of the aligned-allele strings
    compare_prediction  PEPPAN.py:869-901   overlap of every hit with the genome's original annotation (column 10)
    iter_map_bsn        PEPPAN.py:759-867   one genome: search + grouping of merged fragments + allele strings + overlap classes
    get_map_bsn         PEPPAN.py:907-989   all genomes -> the four stores (.tab / .seq / .mat / .conflicts)

The reference runs one forked worker and one uberBlast call per genome (PEPPAN.py:922).  Here `get_map_bsn` hands MANY
genomes to one GPU search (`uberBlastBatch`, pep_set_target_groups keeps the per-genome ranking) and then performs the
same per-genome bookkeeping, in genome order, so the stores are those of the reference's in-order `map` variant
(PEPPAN.py:923).  Everything below the search is host logic pinned by tests/golden/g14_mapbsn.json / g15_getmapbsn.json.
"""
import io
import queue
import threading
import os
import sys
import zipfile

import numpy as np

from .configure import logger

__all__ = ['MapBsn', 'decodeSeq', 'encodeSeq', 'compare_prediction', 'build_bsn', 'iter_map_bsn', 'get_map_bsn']


def _npy_bytes(val):
    buf = io.BytesIO()
    np.lib.format.write_array(buf, np.asanyarray(val), allow_pickle=True)
    return buf.getvalue()


class MapBsn(object):
    """dict-like store: zip member `str(key)` holds one array in .npy format (object arrays pickled).  Readable by
    `np.load(fname, allow_pickle=True)` like the reference's files (PEPPAN.py:1931)."""

    def __init__(self, fname, mode='r'):
        self.fname, self.mode = fname, mode
        self.conn = zipfile.ZipFile(fname, mode=mode, compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1)
        self.namelist = set(self.conn.namelist())
        # writes go through ONE background thread per store, in order: deflate releases the GIL, so the compression of a genome's
        # tables overlaps with the Python bookkeeping of the next one (a third of get_map_bsn's time was spent inside zlib)
        self._queue = self._thread = self._error = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        try:
            self._flush()
        finally:
            if self._thread is not None:
                self._queue.put(None)
                self._thread.join()
                self._thread = None
            self.conn.close()

    def _writer(self):
        while True:
            item = self._queue.get()
            try:
                if item is None:
                    return
                db, key, data = item
                if self._error is None:
                    # members are read back whole either way; deflating a few hundred bytes costs more than it saves (zlib set-up per member)
                    db.writestr(key, data, compress_type=zipfile.ZIP_STORED if len(data) < 4096 else None)
            except BaseException as e:              # reported by the next _flush() on the owning thread
                self._error = e
            finally:
                self._queue.task_done()

    def _flush(self):
        """wait until every queued member is in its archive"""
        if self._thread is not None:
            self._queue.join()
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def exists(self, key):
        return str(key) in self.namelist

    def get(self, key, default=[]):
        key = str(key)
        if key not in self.namelist:
            return default
        self._flush()
        return np.lib.format.read_array(io.BytesIO(self.conn.read(key)), allow_pickle=True)

    __getitem__ = get

    def keys(self):
        return self.namelist

    def values(self):
        for key in self.namelist:
            yield self.get(key)

    def items(self):
        for key in self.namelist:
            yield key, self.get(key)

    def size(self):
        return len(self.namelist)

    def delete(self, key):
        """logical delete: the member stays in the archive but is no longer listed"""
        self.namelist.discard(str(key))

    def pop(self, key, default=[]):
        val = self.get(key, default)
        self.delete(key)
        return val

    def delete_real(self, key):
        """physically drop a member.  The reference shells out to `zip -d` and reopens the archive with mode 'w'
        (PEPPAN.py:71-77), which truncates it; here the archive is rewritten in-process without the member and every other
        member is kept."""
        key = str(key)
        if key not in self.namelist:
            return
        self._flush()
        self.namelist.discard(key)
        self.conn.close()
        tmp = self.fname + '.rewrite'
        with zipfile.ZipFile(self.fname) as src, zipfile.ZipFile(tmp, 'w', compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1) as dst:
            for name in src.namelist():
                if name != key:
                    dst.writestr(name, src.read(name))
        os.replace(tmp, self.fname)
        self.conn = zipfile.ZipFile(self.fname, mode='a', compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1)

    def _save(self, db, key, val):
        data = _npy_bytes(val)                      # serialised here: the caller may change `val` afterwards
        if self._thread is None:
            self._queue = queue.Queue(maxsize=256)
            self._thread = threading.Thread(target=self._writer, daemon=True)
            self._thread.start()
        self._queue.put((db, key, data))

    def save(self, key, val):
        key = str(key)
        self.delete_real(key)
        self._save(self.conn, key, val)
        self.namelist.add(key)

    def update(self, dataset):
        """merge a list of 2-D arrays, each keyed by its [0][0], into the store (rows appended to what the key holds)"""
        tmp_name = self.fname[:-4] + '.tmp.npz'
        seen = set()
        with zipfile.ZipFile(tmp_name, mode='w', compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1) as tmp:
            for d in dataset:
                key = str(d[0][0])
                seen.add(key)
                old = self.get(key)
                self._save(tmp, key, np.vstack([old, d]) if len(old) else d)
            for key in list(self.keys()):
                if key not in seen:
                    data = self.get(key)
                    if len(data):
                        seen.add(key)
                        self._save(tmp, key, data)
            self._flush()
        self.conn.close()
        self.namelist = seen
        os.rename(tmp_name, self.fname)
        self.conn = zipfile.ZipFile(self.fname, mode='a', compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1)


# ------------------------------------------------------------------------------------------------ allele strings
_BASE = np.zeros(256, dtype=np.uint8)
_BASE[[ord(c) for c in 'ACGT']] = (1, 2, 3, 4)


def encodeSeq(b):
    """uint8 base codes (0 gap/unknown, 1..4 ACGT) of length L -> ceil(L/3) bytes: first third * 25 + second third * 5 +
    last third (zero padded), the packing at PEPPAN.py:847-848"""
    s = -(-b.shape[0] // 3)
    tail = np.concatenate([b, np.zeros(-b.shape[0] % 3, dtype=int)])[2 * s:]
    return (b[:s] * 25 + b[s:2 * s] * 5 + tail).astype(np.uint8)


def decodeSeq(seqs):
    """inverse of encodeSeq for a [n, s] matrix (PEPPAN.py:318-324): [n, 3s] codes"""
    n, s = seqs.shape
    out = np.zeros([n, s * 3], dtype=np.uint8)
    out[:, :s] = seqs // 25
    out[:, s:2 * s] = (seqs % 25) // 5
    out[:, 2 * s:] = seqs % 5
    return out


# ------------------------------------------------------------------------------------------------ old annotation
def _stable_order(tab, keys):
    """row order of a stable multi-key sort (first key most significant), like DataFrame.sort_values(by=keys)"""
    cols = []
    for k in reversed(keys):
        c = (k if isinstance(k, np.ndarray) else tab[:, k]).tolist()
        if any(isinstance(v, str) for v in c):
            cols.append(np.unique(np.array(c, dtype=str), return_inverse=True)[1])
        elif all(isinstance(v, (int, np.integer)) for v in c):
            cols.append(np.asarray(c, dtype=np.int64))
        else:
            cols.append(np.asarray(c, dtype=np.float64))
    return np.lexsort(cols)


def _known_fraction(T, old_prediction):
    """Column 10 of the mapping tables (PEPPAN.py:869-901): for every hit the largest fraction of an original gene of the same contig
    that it covers in frame and on the same strand (0.1 when there is none), for all rows of the HitTable at once.  Returns
    (order, known): `order` = the row order in which the reference walks the table, (contig, lower reference coordinate), stable;
    `known[k]` belongs to row order[k].
    An original gene p = [id, start, end, strand, ...] counts for a hit [s, e] when one of its two frame markers is one of the hit's
    and the overlap is >= 0.6 of the gene or of the hit; the hit's markers come from where its query would start / end on the
    reference.  The reference sweeps the contig's genes with a pointer that only moves forward past genes ending before the hit
    and stops at the first gene starting behind it; with the genes in start order (as the store holds them) that is "every
    overlapping gene", which is what the vectorised form evaluates - an unsorted gene list takes the row loop instead."""
    n = len(T)
    lo, hi = np.minimum(T.ss, T.se), np.maximum(T.ss, T.se)
    order = np.lexsort((lo, T.r_codes()))
    known = np.full(n, 0.1, dtype=np.float64)
    if n == 0:
        return order, known
    ri = T.ri[order]
    s, e = lo[order], hi[order]
    fwd = (T.ss < T.se)[order]
    head, tail = (T.ss - T.qs + 1)[order], (T.se + (T.ql - T.qe))[order]
    f1 = np.where(fwd, head % 3 + 1, (-head) % 3 - 1)
    f2 = np.where(fwd, (tail + 1) % 3 + 1, (-(tail - 1)) % 3 - 1)
    bounds = np.concatenate([[0], np.flatnonzero(np.diff(ri)) + 1, [n]])
    with MapBsn(old_prediction) as op:
        for a, b in zip(bounds[:-1].tolist(), bounds[1:].tolist()):
            genes = op.get(T.r_tab[ri[a]])
            if len(genes) == 0:
                continue
            g1 = np.array([p[1] for p in genes], dtype=np.int64)
            g2 = np.array([p[2] for p in genes], dtype=np.int64)
            plus = np.array([p[3] == '+' for p in genes], dtype=bool)
            if np.any(np.diff(g1) < 0):
                known[a:b] = _known_fraction_rows(s[a:b], e[a:b], f1[a:b], f2[a:b], g1, g2, plus)
                continue
            # genes [first, last) can touch the hit: first = the first gene (in store order) that does not end before it
            first = np.searchsorted(np.maximum.accumulate(g2), s[a:b], side='left')
            last = np.maximum(np.searchsorted(g1, e[a:b], side='right'), first)
            cnt = last - first
            tot = int(cnt.sum())
            if tot == 0:
                continue
            row = np.repeat(np.arange(a, b), cnt)
            gi = np.repeat(first - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(tot)
            p1, p2, pl = g1[gi], g2[gi], plus[gi]
            m1 = np.where(pl, p1 % 3 + 1, (-(p1 - 1)) % 3 - 1)
            m2 = np.where(pl, (p2 + 1) % 3 + 1, (-p2) % 3 - 1)
            in_frame = (m1 == f1[row]) | (m1 == f2[row]) | (m2 == f1[row]) | (m2 == f2[row])
            plen = p2 - p1 + 1
            ovl = (np.minimum(e[row], p2) - np.maximum(s[row], p1) + 1).astype(np.float64)
            ok = in_frame & ((ovl >= 0.6 * plen) | (ovl >= 0.6 * (e[row] - s[row] + 1)))
            if ok.any():
                np.maximum.at(known, row[ok], ovl[ok] / plen[ok])
    return order, known


def _known_fraction_rows(s, e, f1, f2, g1, g2, plus):
    """the reference's pointer sweep, row by row (genes not in start order)"""
    out = np.full(len(s), 0.1, dtype=np.float64)
    at, ng = 0, len(g1)
    for k in range(len(s)):
        while at < ng and s[k] > g2[at]:
            at += 1
        for j in range(at, ng):
            if e[k] < g1[j]:
                break
            if plus[j]:
                m1, m2 = g1[j] % 3 + 1, (g2[j] + 1) % 3 + 1
            else:
                m1, m2 = (-(g1[j] - 1)) % 3 - 1, (-g2[j]) % 3 - 1
            if m1 not in (f1[k], f2[k]) and m2 not in (f1[k], f2[k]):
                continue
            plen = g2[j] - g1[j] + 1
            ovl = min(e[k], g2[j]) - max(s[k], g1[j]) + 1.
            if ovl >= 0.6 * plen or ovl >= 0.6 * (e[k] - s[k] + 1):
                out[k] = max(out[k], ovl / plen)
    return out


def _with_known(T, old_prediction):
    """the table in the order compare_prediction returns it - (query, contig, score), stable on top of the (contig, position) walk -
    with column 10 replaced"""
    order, known = _known_fraction(T, old_prediction)
    T = T.take(order)
    T.evalue = known
    return T.take(np.lexsort((T.score, T.r_codes(), T.q_codes())))


def compare_prediction(blastab, old_prediction):
    """column 10 <- the largest fraction of an in-frame, same-strand original gene that a hit covers (0.1 if none);
    returns the table sorted by (query, contig, score).  PEPPAN.py:869-901.  Object rows in and out."""
    from .hittable import HitTable
    if blastab.shape[0] == 0:
        return blastab
    return _with_known(HitTable.from_rows(blastab), old_prediction).to_rows(cigar='str')


# ------------------------------------------------------------------------------------------------ one genome
def _passes_all(length, ql, params):
    return ((length >= np.maximum(params['match_prop'] * ql, params['match_len'])) |
            (length >= np.maximum(params['match_prop1'] * ql, params['match_len1'])) |
            (length >= np.maximum(params['match_prop2'] * ql, params['match_len2'])))


def build_bsn(blastab, overlap, seq, orthoGroup, old_prediction, params, ctx=None):
    """(17-column table with merge groups, int[m,3] overlaps) of ONE genome -> (bsn object[n,7], ovl int[k,3]).
    bsn row = [gene, contig, score, identity, packed allele, group id, rows(object[k,16])].  PEPPAN.py:773-866.
    `blastab` is the HitTable the search chain ends with (the product path: no Python row is touched before the rows that are
    stored get made) or the same thing as object rows.  The per-hit allele strings, their in-frame / stop-free lengths and the
    packing run on the GPU (K12, `ctx.alleles`).

    Groups, in the reference's order: first every row that stands for itself - a row whose merge group is just itself, or a member
    of a chain that also passes the thresholds alone - in table order; then the chains, in the order their first member shows up,
    members in chain order."""
    from .hittable import HitTable
    T = blastab if isinstance(blastab, HitTable) else HitTable.from_rows(blastab)
    if len(T) == 0:
        return np.empty([0, 7], dtype=object), np.zeros([0, 3], dtype=np.int64)
    if ctx is None:
        from .uberBlast import get_context
        ctx = get_context()
    T.q_tab, T.r_tab = [int(x) for x in T.q_tab], [int(x) for x in T.r_tab]
    T.q_sorted = T.r_sorted = False           # (integer names order numerically from here on)
    T = _with_known(T, old_prediction)
    n = len(T)
    n_id = int(T.rid.max()) + 1
    mi = params['match_identity']
    ok = (T.m_span >= 0) & (T.m_iden >= mi) & _passes_all(T.m_span, T.ql, params)         # the row's merge group passes
    kept = np.zeros(n_id, dtype=bool)
    kept[T.rid[ok]] = True
    lone = ok & (T.m_len <= 1)
    also_alone = ok & (T.m_len > 1) & (T.iden >= mi) & _passes_all(T.qe - T.qs + 1, T.ql, params)
    single = np.flatnonzero(lone | also_alone)
    # chains: keyed by their first member's row id, in order of first appearance; every member row finds its slot by its row id
    in_chain = np.flatnonzero(ok & (T.m_len > 1))
    chain_rows, chain_first = [], []
    if len(in_chain):
        key = T.m_ids[T.m_start[in_chain]]
        uniq, first_pos = np.unique(key, return_index=True)
        row_of_id = np.full(n_id, -1, dtype=np.int64)
        row_of_id[T.rid[in_chain]] = in_chain
        rep = in_chain[np.sort(first_pos)]                                         # one member row per chain, same order
        for r in rep.tolist():
            members = T.m_ids[T.m_start[r]:T.m_start[r] + T.m_len[r]]
            rows = row_of_id[members]
            if (rows < 0).any():
                raise ValueError('build_bsn: a chained hit is missing from the table')
            chain_rows.append(rows)
            chain_first.append(r)
    flat = np.concatenate([single] + chain_rows).astype(np.int64) if (len(single) or chain_rows) else np.zeros(0, np.int64)
    n_rows = np.concatenate([np.ones(len(single), dtype=np.int64), np.array([len(r) for r in chain_rows], dtype=np.int64)])
    n_groups = len(n_rows)
    overlap = overlap[kept[overlap.T[0]] & kept[overlap.T[1]], :2]
    if n_groups == 0:
        return np.empty([0, 7], dtype=object), np.zeros([0, 3], dtype=np.int64)
    grp_off = np.concatenate([[0], np.cumsum(n_rows)]).astype(np.uint64)
    first = grp_off[:-1].astype(np.int64)
    head_row = np.concatenate([single, np.array(chain_first, dtype=np.int64)]).astype(np.int64)      # the row a group takes gene / contig / group score from
    is_lone_group = np.concatenate([lone[single], np.zeros(len(chain_rows), dtype=bool)])
    # ---- K12 over every row of every group
    from ._native import LOCUS_DTYPE
    contigs = [sq for nm, sq in seq]
    cidx = {nm: i for i, (nm, sq) in enumerate(seq)}
    loci = np.zeros(len(flat), dtype=LOCUS_DTYPE)
    contig_of = np.array([cidx.get(nm, -1) for nm in T.r_tab], dtype=np.int64)[T.ri[flat]]       # (the name table of a batch search lists every genome's contigs)
    if len(contig_of) and contig_of.min() < 0:
        raise ValueError('build_bsn: a hit names a contig that is not part of this genome')
    loci['contig'] = contig_of
    loci['q_start'], loci['rs'], loci['re'] = T.qs[flat], T.ss[flat], T.se[flat]
    loci['cigar_runs'], loci['cigar_off'] = T.c_runs[flat], T.c_off[flat]
    loci['group'] = np.repeat(np.arange(n_groups), n_rows)
    ql = T.ql[flat]
    in_frame, orf, packed = ctx.alleles(contigs, loci, T.arena, grp_off, ql[first], params['gtable'])
    sc = np.minimum(in_frame, orf + 3)
    iden, known = T.iden[flat], T.evalue[flat]
    q_lo, q_hi = T.qs[flat], T.qe[flat]
    qspan = q_hi - q_lo + 1
    r = np.sqrt(sc.astype(np.float64) / ql * known)
    msc = (sc * iden) * np.sqrt(sc * r)
    amsc = msc / qspan
    pack_off = np.concatenate([[0], np.cumsum((ql[first] + 2) // 3)])
    # ---- assemble
    as_single, as_chain = np.full(n_id, -1, dtype=np.int64), np.full(n_id, -1, dtype=np.int64)
    gid_of_row = loci['group'].astype(np.int64)
    multi = np.repeat(n_rows > 1, n_rows)
    rid_flat = T.rid[flat]
    as_single[rid_flat[~multi]] = gid_of_row[~multi]
    as_chain[rid_flat[multi]] = gid_of_row[multi]
    rows16 = T.take(flat).to_rows(cigar='str')[:, :16]              # the object rows the .mat store keeps, made once
    g_gene, g_contig = [T.q_tab[i] for i in T.qi[head_row].tolist()], [T.r_tab[i] for i in T.ri[head_row].tolist()]
    # group score / identity: the merge group's for a lone row and for a chain, the row's own for a chain member standing alone
    g_iden = np.where(is_lone_group | (n_rows > 1), T.m_iden[head_row], T.iden[head_row]).tolist()
    bsn = np.empty([n_groups, 7], dtype=object)
    lo_l, hi_l = grp_off[:-1].astype(np.int64).tolist(), grp_off[1:].astype(np.int64).tolist()
    for gid in range(n_groups):
        lo, hi = lo_l[gid], hi_l[gid]
        if hi - lo == 1:
            score = msc[lo]
        else:
            spans = [[q_lo[k], q_hi[k], amsc[k], msc[k]] for k in range(lo, hi)]
            for prev, cur in zip(spans[:-1], spans[1:]):          # fragments overlapping on the query: the weaker one is trimmed
                if cur[0] < prev[1]:
                    if cur[2] > prev[2]:
                        prev[1] = cur[0] - 1
                        prev[3] = prev[2] * (prev[1] - prev[0] + 1)
                    else:
                        cur[0] = prev[1] + 1
                        cur[3] = cur[2] * (cur[1] - cur[0] + 1)
            score = np.sum([c[3] for c in spans])
        row = bsn[gid]
        row[0], row[1], row[2], row[3], row[4], row[5], row[6] = g_gene[gid], g_contig[gid], score, g_iden[gid], packed[pack_off[gid]:pack_off[gid + 1]], gid, rows16[lo:hi]
    a0, c0, a1, c1 = as_single[overlap.T[0]], as_chain[overlap.T[0]], as_single[overlap.T[1]], as_chain[overlap.T[1]]
    overlap = np.vstack([np.vstack([m, k]).T[(m >= 0) & (k >= 0)] for m in (a0, c0) for k in (a1, c1)] +
                        [np.vstack([as_single, as_chain]).T[(as_single >= 0) & (as_chain >= 0)]])
    if overlap.shape[0]:
        og = np.load(orthoGroup, allow_pickle=True) if isinstance(orthoGroup, str) else orthoGroup
        rel = {}
        for g in og[og.T[2] != 0]:
            rel[(g[0], g[1])] = 1 if g[2] > 0 else -1
        for g in og[og.T[2] != 0]:
            rel[(g[1], g[0])] = 1 if g[2] > 0 else -1
        score = np.array([0 if m == k else rel.get((m, k), 2) for m, k in zip(bsn[overlap.T[0], 0], bsn[overlap.T[1], 0])], dtype=np.int64)
        overlap = np.hstack([overlap, score[:, np.newaxis]])[score >= 0]
    else:
        overlap = np.zeros([0, 3], dtype=np.int64)
    return bsn, overlap


def _map_argv(clust, params):
    tools = '--blastn' if params.get('noDiamond') else '--blastn --diamond'
    tail = '-t 1 -e 0,3' if params.get('noDiamond') else '-t 1 -s 1 -e 0,3'
    return '-q {0} -f -m -O {1} --min_id {2} --min_cov {3} --min_ratio {4} --merge_gap {5} --merge_diff {6} {7} --gtable {8}'.format(
        clust, tools, params['match_identity'] - 0.1, params['match_frag_len'], params['match_frag_prop'], params['link_gap'],
        params['link_diff'], tail, params['gtable']).split()


def _write_genome(prefix, id, seq):
    gfile = '{0}.{1}.genome'.format(prefix, id)
    with open(gfile, 'w') as fout:
        for n, s in seq:
            fout.write('>{0}\n{1}\n'.format(n, s))
    return gfile


def iter_map_bsn(data):
    """one genome, the reference's worker signature (PEPPAN.py:759-867): writes `<prefix>.<id>.bsn.npz`, returns its prefix"""
    from .uberBlast import uberBlast
    prefix, clust, id, taxon, seq, orthoGroup, old_prediction, params = data
    gfile = _write_genome(prefix, id, seq)
    try:
        blastab, overlap = uberBlast(['-r', gfile] + _map_argv(clust, params))
    finally:
        os.unlink(gfile)
    bsn, ovl = build_bsn(blastab, overlap, seq, orthoGroup, old_prediction, params)
    out_prefix = '{0}.{1}'.format(prefix, id)
    np.savez_compressed(out_prefix + '.bsn.npz', bsn=bsn, ovl=ovl)
    return out_prefix


# ------------------------------------------------------------------------------------------------ all genomes
CHUNK = 1000          # arrays per member of the .seq / .mat stores (PEPPAN.py:953, 962)
BLOCK = 30000         # group ids per member of the .conflicts store (PEPPAN.py:934-947)


def _gpu_search(prefix, clust, jobs, params, genomes_per_batch=64):
    """yield (blastab, overlap) per genome, `genomes_per_batch` genomes per GPU search"""
    from .uberBlast import uberBlastBatch
    argv = _map_argv(clust, params)
    for lo in range(0, len(jobs), genomes_per_batch):
        files = [_write_genome(prefix, id, seq) for id, taxon, seq in jobs[lo:lo + genomes_per_batch]]
        try:
            results = uberBlastBatch(files, argv, as_tables=True)        # (HitTable, overlaps) per genome: build_bsn works on the columns
        finally:
            for f in files:
                os.unlink(f)
        for r in results:
            yield r


def _all_bsn(prefix, clust, jobs, og, old_prediction, params, search, ctx, group, per_round):
    """(job, (bsn, ovl)) for every genome in job order.  With torch.distributed initialised the genomes are dealt to the ranks in
    blocks of `per_round` (independent units, no data-path collective); every rank maps its block on its own GPU and rank 0
    gathers the finished per-genome objects - only rank 0 yields, the others just take part."""
    def local(mine):
        out = []
        for (id, taxon, seq), (blastab, overlap) in zip(mine, search(prefix, clust, mine, params) if mine else ()):
            out.append(build_bsn(blastab, overlap, seq, og, old_prediction, params, ctx))
        return out
    world, rank = 1, 0
    dist = sys.modules.get('torch.distributed')            # only a caller that set up a process group has imported it
    if dist is not None and dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        for job, (blastab, overlap) in zip(jobs, search(prefix, clust, jobs, params)):
            yield job, build_bsn(blastab, overlap, job[2], og, old_prediction, params, ctx)
        return
    n_rounds = -(-len(jobs) // (per_round * world))
    for k in range(n_rounds):
        lo = (k * world + rank) * per_round
        done = local(jobs[lo:lo + per_round])
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(done, gathered, dst=0, group=group)
        if rank == 0:
            for r in range(world):
                lo_r = (k * world + r) * per_round
                for job, out in zip(jobs[lo_r:lo_r + per_round], gathered[r]):
                    yield job, out


def get_map_bsn(prefix, clust, genomes, orthoGroup, old_prediction, conn, seq_conn, mat_conn, clf_conn, saveSeq, params, search=None, ctx=None,
                group=None, genomes_per_round=64):
    """genomes: {contig id: [taxon id, sequence]} -> fills the four MapBsn stores like PEPPAN.py:907-989:
      conn      gene id -> int rows [gene, taxon, score*1e4, ident*1e4, ident*1e4, group id, n fragments], best score first
      seq_conn  chunk no -> object array of packed alleles (only with saveSeq)
      mat_conn  chunk no -> object array of the hit rows of each group
      clf_conn  block no -> CSR [30001 offsets + 30001, partner*10 + class] of group-overlap conflicts
    `search(prefix, clust, jobs, params)` yields (blastab, overlap) per genome in job order (default: batched GPU search).
    Under torch.distributed (one process per GPU) the genomes are sharded over the ranks in blocks of `genomes_per_round`; rank 0
    writes the stores (the other ranks pass None for them), whose contents do not depend on the number of ranks."""
    if len(genomes) == 0:
        raise ValueError('get_map_bsn: no genome to map against')
    taxa = {}
    for g, s in genomes.items():
        taxa.setdefault(s[0], []).append([g, s[1]])
    jobs = [(id, taxon, seq) for id, (taxon, seq) in enumerate(taxa.items())]
    og = np.load(orthoGroup, allow_pickle=True)
    n_group = 0
    seqs, seq_cnt = [], 0
    mats, mat_cnt = [], 0
    tabs, conflicts = [], {}

    def flush_conflicts(block):
        ovl = np.vstack(conflicts.pop(block))
        clf_conn.save(block, np.concatenate([np.cumsum(np.concatenate([[0], np.bincount(ovl.T[0], minlength=BLOCK)])) + BLOCK + 1, ovl.T[1]]))

    per_round = max(1, int(genomes_per_round))
    searcher = search or (lambda *a: _gpu_search(*a, genomes_per_batch=per_round))
    for (id, taxon, seq), (bsn, ovl) in _all_bsn(prefix, clust, jobs, og, old_prediction, params, searcher, ctx, group, per_round):
        bId = id
        last = bId == len(jobs) - 1
        if bsn.shape[0]:
            bsn.T[5] += n_group
            ovl[:, :2] += n_group
            first, n_group = n_group, n_group + bsn.shape[0]
            bsn.T[1] = genomes.get(bsn[0, 1], [-1])[0]
            if ovl.shape[0]:
                for block in np.unique((ovl[:, :2] / BLOCK).astype(int)):
                    conflicts.setdefault(block, [])
                ovl = np.vstack([ovl, ovl[:, (1, 0, 2)]])
                ovl = ovl[np.argsort(ovl.T[0])]
                ovl = np.hstack([(ovl[:, :1] / BLOCK).astype(int), ovl[:, :1] % BLOCK, ovl[:, 1:2] * 10 + ovl[:, 2:]])
                for part in np.split(ovl, np.cumsum(np.unique(ovl.T[0], return_counts=True)[1])[:-1]):
                    conflicts[part[0, 0]].append(part[:, 1:])
                for block in np.arange(int(first / BLOCK), int(n_group / BLOCK)):
                    if block in conflicts:
                        flush_conflicts(block)
            if saveSeq:
                seqs = np.concatenate([seqs, bsn.T[4]])
                pieces = np.split(seqs, np.arange(CHUNK, seqs.shape[0], CHUNK))
                seqs = pieces[-1]
                for s in pieces[:-1]:
                    seq_conn.save(seq_cnt, s)
                    seq_cnt += 1
            bsn.T[4] = bsn.T[3]
            mats = np.concatenate([mats, bsn.T[6]])
            pieces = np.split(mats, np.arange(CHUNK, mats.shape[0], CHUNK))
            mats = pieces[-1]
            for m in pieces[:-1]:
                mat_conn.save(mat_cnt, m)
                mat_cnt += 1
            bsn.T[6] = np.array([len(b) for b in bsn.T[6]], dtype=np.uint8)
            bsn.T[2:5] = bsn.T[2:5] * 10000
            tabs.append(bsn[np.argsort(-bsn.T[2])].astype(int))
        logger('Merged {0}.{1}'.format(prefix, id))
        if (bId % 500 == 499 or last) and len(tabs):
            tab = np.vstack(tabs)
            tab = tab[np.argsort(tab.T[0], kind='mergesort')]
            conn.update(np.split(tab, np.cumsum(np.unique(tab.T[0], return_counts=True)[1])[:-1]))
            tabs = []
    if saveSeq and len(seqs):
        seq_conn.save(seq_cnt, seqs)
    if len(mats):
        mat_conn.save(mat_cnt, mats)
    for block in list(conflicts.keys()):
        flush_conflicts(block)
