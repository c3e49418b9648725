"""The callers either side of the search in PEPPAN.py, with the same inputs, outputs and on-disk artefacts
(SURVEY.md section 8a rows a14-a17), so that the GPU path plugs into an unmodified pipeline run:

    writeGenes(fname, genes, priority)            PEPPAN.py:1023-1039   -> <prefix>.genes + exact-duplicate groups
    iterClust(prefix, genes, geneGroup, params)   PEPPAN.py:1777-1792   -> <prefix>.clust.exemplar/.tab/.npy
    get_similar_pairs(clust, priorities, params)  PEPPAN.py:194-294     -> int[n,3] ortholog / conflict pairs,
                                                                          rewrites the exemplar FASTA, extends clust.npy
    get_gene_group(cluFile, bsnFile)              PEPPAN.py:1590-1609   -> {root: [members in merge order]}
    gene_group_labels(cluFile, bsnFile, n)        same partition as get_gene_group, as labels from the GPU (K10)
"""
import re
from operator import itemgetter

import numpy as np

from . import _native as N
from .clust import getClust
from .configure import logger, readFasta
from .uberBlast import uberBlast, get_context, _as_table


def gene_hashes(seqs, ctx=None):
    """the integer PEPPAN stores per gene and breaks priority ties with: int(hashlib.sha1(seq).hexdigest(), 16)
    (PEPPAN.py:62, 1019), computed for the whole list on the GPU (K13 `pep_sha1`)"""
    ctx = ctx or get_context()
    digests = np.ascontiguousarray(ctx.sha1(seqs))
    from .hittable import _pyrows
    return _pyrows().pep_digest_ints(digests.ctypes.data, len(digests), 20)             # (one C loop; five million of them)


class EdgeTable:
    """The duplicate pairs writeGenes returns - rows [kept gene, duplicate, 10000] - as ONE int64[n, 3] block instead of n Python lists of
    three Python integers (4.7 M rows at BASELINE configs[2]: the lists alone cost four seconds to build and three more to turn back into
    the array iterClust saves).  It answers what PEPPAN does with the value (PEPPAN.py:1880-1884, 1790-1791): append() of further rows,
    np.array(table, dtype=int), len(), iteration, indexing, comparison with a list of rows; tolist() gives the plain list."""

    def __init__(self, block):
        self._parts = [np.ascontiguousarray(block, dtype=np.int64).reshape(-1, 3)]          # int64 blocks and plain lists of rows, in order

    def append(self, row):
        if not isinstance(self._parts[-1], list):
            self._parts.append([])
        self._parts[-1].append(row)

    def extend(self, rows):
        if not isinstance(self._parts[-1], list):
            self._parts.append([])
        self._parts[-1].extend(rows)

    def extend_block(self, block):
        """rows as an int64[m, 3] block (iterClust's edges of one identity level)"""
        self._parts.append(np.ascontiguousarray(block, dtype=np.int64).reshape(-1, 3))

    def __len__(self):
        return sum(len(p) for p in self._parts)

    def __iter__(self):
        for p in self._parts:
            yield from (p if isinstance(p, list) else p.tolist())

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.tolist()[i]
        if i < 0:
            i += len(self)
        for p in self._parts:
            if i < len(p):
                return p[i] if isinstance(p, list) else p[i].tolist()
            i -= len(p)
        raise IndexError('EdgeTable index out of range')

    def __array__(self, dtype=None, copy=None):
        parts = [p if not isinstance(p, list) else np.array(p, dtype=dtype or np.int64).reshape(-1, 3) for p in self._parts if len(p)]
        out = parts[0] if len(parts) == 1 else (np.vstack(parts) if parts else self._parts[0])
        return out.astype(dtype) if dtype is not None and out.dtype != np.dtype(dtype) else (out.copy() if copy else out)

    def tolist(self):
        out = []
        for p in self._parts:
            out += p if isinstance(p, list) else p.tolist()
        return out

    def __eq__(self, other):
        if isinstance(other, EdgeTable):
            other = other.tolist()
        return self.tolist() == other

    def __repr__(self):
        return 'EdgeTable(%d rows)' % len(self)


def _lexsort_priority(a0, a1, code):
    """order of the rows under (a0, a1, 160-bit code as 20 big-endian bytes), stable: what sorted(..., key=itemgetter(1)) gives for PEPPAN's
    priorities [file rank, -length, sha1 code] (PEPPAN.py:746-751, 1027)"""
    h0 = np.ascontiguousarray(code[:, 0:8]).view('>u8').ravel().astype(np.uint64)
    h1 = np.ascontiguousarray(code[:, 8:16]).view('>u8').ravel().astype(np.uint64)
    h2 = np.ascontiguousarray(code[:, 16:20]).view('>u4').ravel().astype(np.uint64)
    keys = [h2, h1, h0]
    # the two small integers in one column when their ranges allow it (one stable pass less over five million rows)
    lo0, lo1 = int(a0.min()), int(a1.min())
    r0, r1 = int(a0.max()) - lo0 + 1, int(a1.max()) - lo1 + 1
    if r0 * r1 < (1 << 62):
        keys.append((a0 - lo0) * r1 + (a1 - lo1))
    else:
        keys += [a1, a0]
    return np.lexsort(keys)


def _priority_order(names, priority):
    """the names in the order sorted(priority.items(), key=itemgetter(1)) visits them (PEPPAN.py:1027): PEPPAN's priorities are
    [file rank, -length, sha1 code] (PEPPAN.py:746-751) - three integers, the last one of 160 bits - and a stable lexsort over the
    columns (the code split into 64 + 64 + 32 bits) orders five million of them in seconds where sorting the Python lists takes most of a
    minute.  Anything else (other types, other lengths) is sorted the plain way."""
    vals = [priority[n] for n in names]
    m = len(vals)
    try:
        if m == 0 or any(len(v) != 3 for v in vals):
            raise TypeError
        a0 = np.fromiter((v[0] for v in vals), dtype=np.int64, count=m)
        a1 = np.fromiter((v[1] for v in vals), dtype=np.int64, count=m)
        if any(type(v[0]) is not int or type(v[1]) is not int for v in vals[:64]):
            raise TypeError
        code = np.frombuffer(b''.join([int(v[2]).to_bytes(20, 'big') for v in vals]), dtype=np.uint8).reshape(m, 20)
        return _lexsort_priority(a0, a1, code)
    except (TypeError, ValueError, OverflowError, AttributeError):
        return np.array(sorted(range(m), key=lambda i: vals[i]), dtype=np.int64)


def _scan_genes(genes, priority):
    """names (dictionary order of `priority`, genes with a non-empty sequence only), their priority order, sequence lengths and sha1 digests
    in ONE C pass over the two dictionaries (csrc/pyrows.c pep_genes_scan) -> (names, order index, lengths uint32, digests uint8[n, 20]),
    or None when the values are not PEPPAN's [int, int, 160-bit int] / plain dicts (the caller then goes the Python way)"""
    import ctypes as C
    from .hittable import _pyrows
    if type(genes) is not dict or type(priority) is not dict:
        return None
    cap = len(priority)
    p0, p1, seq_len = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.int64)
    code, digest = np.empty((cap, 20), np.uint8), np.empty((cap, 20), np.uint8)
    names = []
    n = _pyrows().pep_genes_scan(priority, genes, names, *[C.c_void_p(a.ctypes.data) for a in (p0, p1, code, seq_len, digest)], cap)
    if n < 0:
        return None
    if n == 0:
        return names, np.zeros(0, np.int64), np.zeros(0, np.uint32), np.zeros((0, 20), np.uint8)
    idx = _lexsort_priority(p0[:n], p1[:n], code[:n])
    return names, idx, seq_len[:n][idx].astype(np.uint32), np.ascontiguousarray(digest[:n][idx])


def writeGenes(fname, genes, priority, ctx=None):
    """genes in priority order; a gene whose (length, sha1) equals an already written one of the SAME length run is
    reported as its duplicate.  The reference rebuilds its seen-table whenever a length not currently in it shows up
    (PEPPAN.py:1032-1033), so duplicates are only found while one length is 'open'.  The collapse itself runs on the GPU
    (K13 `pep_dedup`: smallest priority index per (length run, digest)); the two dictionaries are read in one C pass (names, priorities,
    lengths, digests as columns), the priority order is a lexsort over those columns and the FASTA leaves in one write."""
    ctx = ctx or get_context()
    scanned = _scan_genes(genes, priority)
    if scanned is not None:
        names, idx, lengths, digests = scanned
        order = np.array(names, dtype=object)[idx] if len(names) else np.zeros(0, dtype=object)
    else:
        names = [n for n in priority if n in genes and len(genes[n][6])]          # dictionary order: ties of the sort keep it (sorted() is stable)
        idx = _priority_order(names, priority)
        order = np.array([names[i] for i in idx.tolist()], dtype=object) if len(names) else np.zeros(0, dtype=object)
        lengths = np.fromiter((len(genes[n][6]) for n in order), dtype=np.int64, count=len(order)).astype(np.uint32)
        digests = np.frombuffer(b''.join([int(genes[n][5]).to_bytes(20, 'big') for n in order]), dtype=np.uint8).reshape(-1, 20)
    rep = ctx.dedup(lengths, digests).astype(np.int64)
    own = rep == np.arange(len(rep))
    with open(fname, 'w') as fout:
        fout.write(''.join(['>{0}\n{1}\n'.format(n, genes[n][6]) for n in order[own].tolist()]))
    dup = np.flatnonzero(~own)
    head = order[:64].tolist() + order[dup[:64]].tolist()
    if len(dup) and all(type(n) is int for n in head):
        try:
            ids = order.astype(np.int64)
            pairs = np.column_stack([ids[rep[dup]], ids[dup], np.full(len(dup), 10000, dtype=np.int64)])
            return fname, (EdgeTable(pairs) if len(pairs) > 100000 else pairs.tolist())      # (small tables stay plain lists)
        except (OverflowError, TypeError, ValueError):
            pass
    return fname, [[a, b, 10000] for a, b in zip(order[rep[dup]].tolist(), order[dup].tolist())]


def identity_schedule(target):
    """the identity levels of the iterative clustering: 1.00, 0.99, ... down to the target (11 levels for 0.90)"""
    return [float(x) for x in np.round(np.arange(1., target - 0.005, -0.01), 5)]


_TWO_INT_COLUMNS = re.compile(r'(?:[+-]?\d+\t[+-]?\d+\n)*')


def _tab_pairs_after_first_line(tab):
    """(gene, exemplar) rows of a clust.tab file WITHOUT its first line, columns typed the way a header-inferring CSV reader
    types them (all-integer column -> ints).  The reference reads the file with pandas' default header handling
    (PEPPAN.py:1786), so the first gene of every table silently never reaches clust.npy; downstream results depend on it.
    -> int64[n, 2] when both columns are integers that fit (what PEPPAN's encoded gene names give), else a list of tuples."""
    with open(tab) as fin:
        text = fin.read()
    if _TWO_INT_COLUMNS.fullmatch(text):                   # one C pass instead of a regular expression per cell (a million of them per level)
        try:
            return np.array(text.split()).astype(np.int64).reshape(-1, 2)[1:]
        except (OverflowError, ValueError):
            pass
    rows = [line.split('\t')[:2] for line in text.split('\n')[:-1 if text.endswith('\n') else None]][1:]
    cols = [[r[0] for r in rows], [r[1] for r in rows]]
    for c in cols:
        if c and all(re.fullmatch(r'[+-]?\d+', x) for x in c):
            c[:] = [int(x) for x in c]
    return list(zip(*cols))


def _canonical_ints(names):
    """the integers n with str(n) among the names"""
    out = []
    for x in names:
        try:
            v = int(x)
        except ValueError:
            continue
        if str(v) == x and -(1 << 63) <= v < (1 << 63):
            out.append(v)
    return np.array(out, dtype=np.int64)


def iterClust(prefix, genes, geneGroup, params):
    """clustering at falling identity levels, each level on the exemplars of the one before; appends [exemplar, member,
    identity label] edges to geneGroup, saves them as <prefix>.clust.npy and returns the last exemplar file"""
    coverage = np.round(params['coverage'], 2)
    current = genes
    for level in identity_schedule(params['identity']):
        params.update(identity=level, coverage=coverage)
        current, tab = getClust(prefix, current, params)
        exemplars = readFasta(current, headOnly=True)
        logger('Iterative clustering. {0} exemplars left with identity = {1}'.format(len(exemplars), level))
        label = int(min(1., level + 0.005) * 10000)
        pairs = _tab_pairs_after_first_line(tab)
        if isinstance(pairs, np.ndarray):                  # integer names: the level's edges as one block
            pairs = pairs[pairs[:, 0] != pairs[:, 1]]
            first = np.isin(pairs[:, 0], _canonical_ints(exemplars))
            block = np.column_stack([np.where(first, pairs[:, 0], pairs[:, 1]), np.where(first, pairs[:, 1], pairs[:, 0]),
                                     np.full(len(pairs), label, dtype=np.int64)])
            if hasattr(geneGroup, 'extend_block'):
                geneGroup.extend_block(block)
            else:
                geneGroup.extend(block.tolist())
            continue
        for gene, other in pairs:
            if gene != other:
                geneGroup.append([gene, other, label] if str(gene) in exemplars else [other, gene, label])
    np.save('{0}.clust.npy'.format(prefix), np.array(geneGroup, dtype=int))
    return current


# ---- the decision pass over the all-vs-all table (PEPPAN.py:194-294; behaviour spec: SURVEY.md appendix A.4) -----------------
def _classify_rows(T, rank_q, rank_r, q, r, near_identity, cover):
    """the row-local tests of PEPPAN.py:244-263 for all rows of the numeric table at once -> (action code per row, forward flag per row,
    int(identity * 10000) per row).  Arithmetic in float64 exactly as the reference's float() conversions make it."""
    f = lambda a: a.astype(np.float64)
    iden, qs, qe, ss, se, ql, sl = T.iden, f(T.qs), f(T.qe), f(T.ss), f(T.se), f(T.ql), f(T.sl)
    q_span, r_span = qe - qs + 1, np.abs(se - ss) + 1
    near = (q != r) & (iden >= near_identity)
    same_head, same_tail = (qs % 3 == ss % 3), ((ql - qe) % 3 == (sl - se) % 3)
    off_frame = (ss > se) | (~same_head & same_tail)                              # reverse strand, or shifted at the head only
    in_frame = ~off_frame & (ss < se) & same_head & same_tail
    root = np.sqrt(cover)
    action = np.full(len(T), N.ROW_ORDINARY, dtype=np.uint8)
    action[near & off_frame & ((q_span >= cover * ql) | (r_span >= cover * sl))] = N.ROW_CONFLICT
    action[near & in_frame & (ql <= sl) & (q_span >= root * sl) & (rank_q >= rank_r)] = N.ROW_ABSORB_QUERY
    action[near & in_frame & (ql > sl) & (r_span >= root * ql) & (rank_q <= rank_r)] = N.ROW_ABSORB_REF
    return action, (ss < se).astype(np.uint8), (iden * 10000.).astype(np.int64).astype(np.int32)


def _self_search(clust, params, pool):
    tools = '--blastn' if params['noDiamond'] else '--blastn --diamond -s 1'
    argv = ['-r', clust, '-q', clust] + tools.split() + ['--min_id', str(params['match_identity'] - 0.05), '--min_cov', str(params['match_frag_len']),
                                                        '-t', str(params['n_thread']), '--min_ratio', str(params['match_frag_prop']), '-e', '3,3', '-p',
                                                        '--gtable', str(params['gtable'])]
    if params.get('sensitive'):                    # (not a PEPPAN parameter: four seed shapes in the translated search, see uberBlast --sensitive)
        argv.append('--sensitive')
    return uberBlast(argv, pool, as_table=True)


def get_similar_pairs(clust, priorities, params, pool=None, ctx=None, timing=None):
    """All-vs-all search of the exemplars, then the reference's ONE ordered pass over its table (rows sorted by query, reference, score with
    the names compared as strings) deciding which exemplars are absorbed by a near-identical in-frame partner, which pairs conflict (-2)
    and which are ortholog-like (mean identity * 1e4).  Nothing here walks rows in Python and no row ever becomes a Python object:
      * the search hands over its numeric HitTable (columns + CIGAR arena);
      * the row-local tests are one pass of host C++ over the columns (pep_similar_classify; _classify_rows is their numpy statement, kept as the
        test's yardstick);
      * the order-dependent state - alive[g] (0 once g was absorbed or found repetitive; rows touching a dead gene are ignored from then
        on) and the pending forward rows of the current (query, reference) pair, settled when the NEXT surviving row belongs to another
        pair - is one pass of host C++ inside the library (pep_similar_scan); it does not depend on what get_similar returns, so
      * get_similar itself (PEPPAN.py:195-224) runs afterwards for all settled pairs at once on the GPU (K14, pep_pair_support), and
      * pep_similar_resolve replays the dictionary writes in order.
    Side effects as in the reference: the exemplar FASTA loses the dead genes, absorbed pairs are added to clust.npy."""
    import time
    t0 = time.perf_counter()
    T = _as_table(_self_search(clust, params, pool))
    t1 = time.perf_counter()
    pairs = np.array([], dtype=int)
    alive_ids, absorbed = set(), []
    marks = [('search', t1)]
    mark = lambda what: marks.append((what, time.perf_counter()))
    if len(T):
        def as_ids(tab):                                                                # the reference casts both name columns to int (PEPPAN.py:231): int() of every name
            return np.fromiter(map(int, tab), dtype=np.int64, count=len(tab))           # (a quarter of the time of numpy's own text -> integer conversion of a str array)
        q_ids = as_ids(T.q_tab)
        r_ids = q_ids if T.r_tab is T.q_tab else as_ids(T.r_tab)
        mark('ids')
        seen_q, seen_r = np.zeros(len(q_ids), dtype=bool), np.zeros(len(r_ids), dtype=bool)
        seen_q[T.qi] = True
        seen_r[T.ri] = True
        genes = np.unique(np.concatenate([q_ids[seen_q], r_ids[seen_r]]))      # sorted: codes keep the order of the ids
        q, r = np.searchsorted(genes, q_ids)[T.qi], np.searchsorted(genes, r_ids)[T.ri]
        rank = np.array(list(map(itemgetter(0), map(priorities.__getitem__, genes.tolist()))))
        rank_q, rank_r = rank[q], rank[r]
        mark('codes')
        action, forward, iden4 = N.similar_classify(T, q, r, rank_q >= rank_r, rank_q <= rank_r, params['clust_identity'], params['clust_match_prop'])
        mark('classify')
        sc = N.similar_scan(q, r, action, forward, iden4, len(genes))
        mark('scan')
        rows_of = sc['ev_rows']
        sup = np.zeros(len(rows_of), dtype=N.SUPPORT_ROW_DTYPE)
        sup['q_start'], sup['r_start'], sup['identity'] = T.qs[rows_of], T.ss[rows_of], T.iden[rows_of]
        sup['cigar_off'], sup['cigar_runs'] = T.c_off[rows_of], T.c_runs[rows_of]
        off = sc['ev_row_off']
        # lengths of a group's two genes from its first row (conflict events carry no rows: any row will do, they are not judged)
        first = rows_of[np.minimum(off[:-1], len(rows_of) - 1)] if len(rows_of) else np.zeros(len(off) - 1, dtype=np.int64)
        value = (ctx or get_context()).pair_support(sup, T.arena, off, T.ql[first], T.sl[first], N.support_limits(params))
        mark('pair_support')
        res = N.similar_resolve(sc['ev_kind'], sc['ev_a'], sc['ev_b'], value)
        if len(res):
            pairs = np.column_stack([genes[res[:, 0]], genes[res[:, 1]], res[:, 2]]).astype(int)
        alive_ids = set(genes[(sc['alive'] > 0) & (sc['seen_as_query'] > 0)].tolist())
        ab = sc['absorbed']
        absorbed = np.column_stack([genes[ab[:, 0]], genes[ab[:, 1]], ab[:, 2]]).tolist() if len(ab) else []
        mark('resolve')
    _drop_dead_exemplars(params['clust'], alive_ids)
    mark('rewrite')
    if absorbed:
        npy = params['clust'].rsplit('.', 1)[0] + '.npy'
        edges = np.vstack([np.load(npy, allow_pickle=True), absorbed])
        np.save(npy, edges[np.argsort(-edges.T[2])])
    if timing is not None:
        timing.update(search_ms=(t1 - t0) * 1e3, decide_ms=(time.perf_counter() - t1) * 1e3, rows=len(T),
                      decide_parts_ms={b[0]: (b[1] - a[1]) * 1e3 for a, b in zip(marks, marks[1:])})
    return pairs


def _drop_dead_exemplars(fasta, alive_ids):
    """rewrite the exemplar FASTA in place, keeping the records (header line and the lines behind it, byte for byte) of genes that appeared as
    a query and are still alive (PEPPAN.py:278-288).  The header lines are located with bytes.find over the file's buffer (a regular
    expression over the 10 MB of 10 000 exemplars took twelve times as long), the names are tested against the alive set as one integer
    column, and the kept stretches are written straight from that buffer; the file is left alone when every record stays."""
    try:
        ids = np.fromiter(alive_ids, dtype=np.int64, count=len(alive_ids))
        if N.fasta_keep(fasta, ids) is not None:                  # host C++ (pep_fasta_keep): one read, one write
            return
    except (OverflowError, TypeError, ValueError):
        pass                                                      # names that are not machine integers: the Python way below
    with open(fasta, 'rb') as fin:
        data = fin.read()
    find = data.find
    starts = [0] if data[:1] == b'>' else []
    p = find(b'\n>')
    while p >= 0:
        starts.append(p + 1)
        p = find(b'\n>', p + 2)
    n = len(starts)
    heads = [data[s + 1:s + 65] for s in starts]                  # enough for the name: the first token of the header line
    try:
        names = [h.split(None, 1)[0] if h[:1] not in b' \t\r\n' else b'' for h in heads]
        if any(len(h) == 64 and len(t) == 64 for h, t in zip(heads, names)):
            raise ValueError                                       # a name longer than the window: the plain way below
        ids = np.array(names, dtype='S64').astype(np.int64) if n else np.zeros(0, np.int64)
        keep = np.isin(ids, np.fromiter(alive_ids, dtype=np.int64, count=len(alive_ids)))
    except (ValueError, IndexError, OverflowError):
        keep = np.zeros(n, dtype=bool)
        for k, s in enumerate(starts):
            e = find(b'\n', s)
            name = data[s + 1:e if e >= 0 else len(data)].split()
            keep[k] = bool(name) and int(name[0]) in alive_ids
    if keep.all() and (not starts or starts[0] == 0):
        return
    starts.append(len(data))
    view = memoryview(data)
    # runs of kept records, each written in one piece
    edge = np.flatnonzero(np.diff(np.concatenate([[False], keep, [False]]).astype(np.int8)))
    with open(fasta, 'wb') as fout:                                # (one write: a call per run of records cost more than the copy)
        fout.write(b''.join([view[starts[k]:starts[j]] for k, j in zip(edge[0::2].tolist(), edge[1::2].tolist())]))


def _edges(cluFile, bsnFile):
    clu = np.load(cluFile.rsplit('.', 1)[0] + '.npy', allow_pickle=True)
    bsn = np.load(bsnFile, allow_pickle=True)
    return clu, bsn[bsn.T[2] > 0]


def get_gene_group(cluFile, bsnFile):
    """single-linkage groups: the q side's whole group is appended to the r side's group and re-tagged with its root
    (merge order is kept - downstream code depends on it)"""
    groups, tag = {}, {}
    for matrix in _edges(cluFile, bsnFile):
        for r, q, _ in matrix:
            q_root, r_root = tag.get(q, q), tag.get(r, r)
            if q_root == r_root:
                continue
            moved = groups.pop(q_root, [q_root])
            for x in moved:
                tag[x] = r_root
            groups[r_root] = groups.get(r_root, [r_root]) + moved
    return groups


def gene_group_labels(cluFile, bsnFile, n_genes=None, device=None):
    """the same partition as get_gene_group, computed on the GPU (K10): label[g] = smallest gene id of g's group"""
    clu, bsn = _edges(cluFile, bsnFile)
    e = np.vstack([clu[:, :2], bsn[:, :2]]).astype(np.int64) if len(clu) + len(bsn) else np.zeros((0, 2), np.int64)
    n = int(n_genes if n_genes is not None else (e.max() + 1 if len(e) else 0))
    return get_context(device).components(n, e[:, 0], e[:, 1])
