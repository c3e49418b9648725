"""CPU tests of the host-side helpers behind the pipeline entry points (no GPU): the C passes over Python objects (csrc/pyrows.c), the
exemplar rewrite (pep_fasta_keep, host C++ inside the library) and the block form of writeGenes' duplicate table."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_fasta_keep_equals_the_python_rewrite(tmp_path):
    """pep_fasta_keep (PEPPAN.py:278-288): records kept verbatim, text before the first header dropped, untouched file when all stay,
    names that are not plain integers handed back to the Python path, which applies int()"""
    from peppan_amd import _native as N, pipeline as PL
    rng = np.random.default_rng(3)
    recs = []
    for i in range(300):
        body = ''.join('ACGT'[x] for x in rng.integers(0, 4, int(rng.integers(1, 200))))
        recs.append('>%d%s\n%s\n' % (i * 3 - 50, ' some description' if i % 5 == 0 else '', '\n'.join(body[k:k + 60] for k in range(0, len(body), 60))))
    text = 'stray line\n' + ''.join(recs) + '>9999'                      # a last record without a newline
    alive = {i * 3 - 50 for i in range(300) if i % 4} | {9999, 123456}
    a, b = tmp_path / 'a.fa', tmp_path / 'b.fa'
    a.write_text(text); b.write_text(text)
    assert N.fasta_keep(str(a), sorted(alive)) == (301, len([i for i in range(300) if i % 4]) + 1)
    keep = N.fasta_keep
    try:
        N.fasta_keep = lambda path, ids: None                            # force the Python way
        PL._drop_dead_exemplars(str(b), alive)
    finally:
        N.fasta_keep = keep
    assert a.read_bytes() == b.read_bytes() and a.read_text().startswith('>-47\n') and a.read_text().endswith('>9999')
    # every record stays and the file starts with a header: not rewritten
    c = tmp_path / 'c.fa'
    c.write_text('>1\nAC\n>2\nGT\n')
    before = os.stat(str(c)).st_mtime_ns
    assert N.fasta_keep(str(c), [1, 2, 3]) == (2, 2) and os.stat(str(c)).st_mtime_ns == before
    # a name that is not a plain decimal integer: no change, None -> the caller's own rules (int() accepts 1_0, refuses x7)
    c.write_text('>1_0\nAC\n>2\nGT\n')
    assert N.fasta_keep(str(c), [10]) is None and c.read_text() == '>1_0\nAC\n>2\nGT\n'
    PL._drop_dead_exemplars(str(c), {10})
    assert c.read_text() == '>1_0\nAC\n'
    c.write_text('>x7\nAC\n')
    with pytest.raises(ValueError):
        PL._drop_dead_exemplars(str(c), {7})
    assert N.fasta_keep(str(tmp_path / 'missing.fa'), [1]) is None
    c.write_text('')
    assert N.fasta_keep(str(c), [1]) == (0, 0)


def test_genes_scan_equals_the_python_front_end(tmp_path):
    """writeGenes through the C pass over the two dictionaries (pep_genes_scan) == the Python comprehensions it replaces: file, duplicate
    pairs; dictionary order kept for ties, empty sequences and names without a gene skipped, odd priority values fall back"""
    from peppan_amd import pipeline as PL, synth
    from oracle_context import OracleContext
    seqs = synth.make_instances(300, 20, seed=5)
    n = len(seqs)
    code = [int(hashlib.sha1(s.encode()).hexdigest(), 16) for s in seqs]
    genes = {i: ['f', '', 0, 0, '+', code[i], seqs[i]] for i in range(n)}
    prio = {i: [i % 3, -len(seqs[i]), code[i]] for i in reversed(range(n))}           # dictionary order != key order
    genes[7][6] = ''
    del genes[11]
    prio[n + 5] = [0, -10, 12345]                                                     # no such gene
    ctx = OracleContext()
    assert PL._scan_genes(genes, prio) is not None
    fn1, g1 = PL.writeGenes(str(tmp_path / 'c.genes'), genes, prio, ctx=ctx)
    scan = PL._scan_genes
    try:
        PL._scan_genes = lambda g, p: None
        fn2, g2 = PL.writeGenes(str(tmp_path / 'py.genes'), genes, prio, ctx=ctx)
    finally:
        PL._scan_genes = scan
    assert open(fn1).read() == open(fn2).read() and g1 == g2 and len(g1) > 1000
    # values the C pass does not take: it says so and the Python path decides
    assert PL._scan_genes(genes, {**prio, 3: [0.5, -3, 1]}) is None
    assert PL._scan_genes(genes, {**prio, 3: [0, -3, -1]}) is None
    assert PL._scan_genes(genes, {**prio, 3: [0, -3, 1 << 161]}) is None
    assert PL._scan_genes({}, {}) is not None and len(PL._scan_genes({}, {})[0]) == 0


def test_edge_table_behaves_like_the_list_peppan_expects():
    from peppan_amd.pipeline import EdgeTable
    rows = np.arange(30, dtype=np.int64).reshape(10, 3)
    e = EdgeTable(rows)
    assert len(e) == 10 and e[2] == [6, 7, 8] and e[-1] == [27, 28, 29] and e == rows.tolist() and list(e) == rows.tolist()
    e.append([100, 101, 9950])                                        # iterClust (PEPPAN.py:1790)
    e.extend([[200, 201, 9900]])
    a = np.array(e, dtype=int)                                        # PEPPAN.py:1791
    assert a.shape == (12, 3) and a.dtype == np.dtype(int) and a[-2].tolist() == [100, 101, 9950] and e[10] == [100, 101, 9950]
    assert e.tolist() == rows.tolist() + [[100, 101, 9950], [200, 201, 9900]] and e[1:3] == rows[1:3].tolist()
    assert np.array(EdgeTable(np.zeros((0, 3), dtype=np.int64)), dtype=int).shape == (0, 3)


def test_fasta_scan_equals_the_python_reader():
    """pep_fasta_scan (the clusterer's input in one C pass) against clust.blocks_of / sequence_of + the code table, on the shapes a FASTA
    file takes: leading junk, comment lines, blank lines, CRLF, lower case, a header without body, no final newline, '>' inside a line"""
    from peppan_amd import _native as N, clust as CL, linclust as LC
    rng = np.random.default_rng(5)
    texts = [
        '',
        'no header at all\nACGT\n',
        'junk line\n>a desc\nACGT\nacgtn\n>b\n\n# a comment\nAC GT\tA\r\n#>not a header\n>c\n>d x y\nTTTT',
        '>only',
        '>x\nAC>GT\n>y\nA\x1cC\x0bG\n',
        '\n>late\nACGT\n',
        '>p1\nMKV LAW\n>p2\nmkvlaw*\n',
    ]
    recs = []
    for i in range(300):
        seq = ''.join(rng.choice(list('ACGTacgtN-'), size=int(rng.integers(0, 200))))
        lines = [seq[j:j + 60] for j in range(0, len(seq), 60)]
        if i % 7 == 0:
            lines.insert(len(lines) // 2, '# remark')
        recs.append('>g%d some text\n%s%s' % (i, '\n'.join(lines), '\n' if lines else ''))
    texts.append(''.join(recs))
    for text in texts:
        blocks = CL.blocks_of(text)
        for table, protein in ((LC._NT, False), (LC._AA, True)):
            want = [LC.encode(CL.sequence_of(b), protein) for b in blocks]
            got = N.fasta_scan(text.encode('ascii'), table, len(blocks))
            assert got is not None, text[:40]
            codes, off = got
            assert len(off) == len(blocks) + 1 and off[0] == 0
            for k, w in enumerate(want):
                assert np.array_equal(codes[int(off[k]):int(off[k + 1])], w), (text[:40], k)
            assert int(off[-1]) == sum(map(len, want))
        # a record count that does not fit the text is refused, not truncated
        assert N.fasta_scan(text.encode('ascii'), LC._NT, len(blocks) + 1) is None
        if blocks:
            assert N.fasta_scan(text.encode('ascii'), LC._NT, len(blocks) - 1) is None
    # non-ASCII sequence bytes: the caller's own rules apply
    assert N.fasta_scan('>a\nACéT\n'.encode('utf-8'), LC._NT, 1) is None


def test_edge_table_blocks_and_the_tab_reader(tmp_path):
    """EdgeTable.extend_block keeps the order of appends; the one-pass reader of clust.tab gives what the per-cell reader gives"""
    from peppan_amd import pipeline as PL
    e = PL.EdgeTable(np.array([[1, 2, 10000], [3, 4, 10000]]))
    e.append([5, 6, 9900])
    e.extend_block(np.array([[7, 8, 9800], [9, 10, 9800]]))
    e.append([11, 12, 9700])
    want = [[1, 2, 10000], [3, 4, 10000], [5, 6, 9900], [7, 8, 9800], [9, 10, 9800], [11, 12, 9700]]
    assert e.tolist() == want and list(e) == want and len(e) == 6 and e == want
    assert [e[i] for i in range(-6, 6)] == want + want and e[1:3] == want[1:3]
    assert np.array_equal(np.array(e, dtype=int), np.array(want))
    with pytest.raises(IndexError):
        e[6]
    tab = tmp_path / 'x.clust.tab'
    for text, typed in (('1\t1\n10\t1\n-3\t+7\n', np.ndarray), ('1\t1\n10\ta\n3\t7\n', list), ('1\t1\n', np.ndarray), ('', np.ndarray),
                        ('1\t1\n99999999999999999999\t1\n', list), ('1\t1\n2\t1', list), ('a\tb\nc\td\n', list)):
        tab.write_text(text)
        got = PL._tab_pairs_after_first_line(str(tab))
        assert isinstance(got, typed), text
        rows = [line.split('\t')[:2] for line in text.splitlines()][1:]
        cols = [[r[0] for r in rows], [r[1] for r in rows]]
        for c in cols:
            if c and all(x.lstrip('+-').isdigit() for x in c):
                c[:] = [int(x) for x in c]
        assert [tuple(r) for r in (got.tolist() if isinstance(got, np.ndarray) else got)] == list(zip(*cols)), text
    assert PL._canonical_ints(['12', '012', 'x', '-4', '+5', '7 ']).tolist() == [12, -4]


def test_packing_of_big_lists_and_digest_integers():
    """the C loops behind _pack (lists of >= 4096 str / bytes) and gene_hashes' integers against their Python statements"""
    from peppan_amd import _native as N
    from peppan_amd.hittable import _pyrows
    rng = np.random.default_rng(11)
    seqs = [''.join(rng.choice(list('ACGTN'), size=int(rng.integers(0, 40)))) for _ in range(5000)]
    for lst in (seqs, [s.encode() for s in seqs], [s if i % 2 else s.encode() for i, s in enumerate(seqs)]):
        res, off = N._pack(lst)
        joined = b''.join(s if isinstance(s, bytes) else s.encode() for s in lst)
        assert res[:len(joined)].tobytes() == joined and int(off[-1]) == len(joined)
        assert np.array_equal(np.diff(off.astype(np.int64)), [len(s) for s in lst])
    with pytest.raises(UnicodeEncodeError):
        N._pack(seqs[:4999] + ['ACé'])
    res, off = N._pack([''] * 5000)
    assert int(off[-1]) == 0 and res.size >= 1
    d = rng.integers(0, 256, (1000, 20), dtype=np.uint8)
    d[0] = 0
    d[1] = 255
    assert _pyrows().pep_digest_ints(d.ctypes.data, len(d), 20) == [int.from_bytes(x.tobytes(), 'big') for x in d]
    assert _pyrows().pep_digest_ints(d.ctypes.data, 0, 20) == []


def test_read_fasta_through_the_library_equals_the_record_by_record_reader(tmp_path):
    """configure.readFasta / readFastq cut a plain-ASCII file with one pass of host C++ (pep_fasta_records) and everything else record by record:
    both give the reference's dictionary (configure.py:118-128: first word of the header, body lines not starting with '#', white space dropped,
    upper case, the later of two records with one name) - on made-up files with every blank str.split knows, '#' and '>' inside lines, junk in
    front of the first header, headers without a body, a last line without its newline, duplicate names"""
    import gzip
    import random
    from peppan_amd import configure as CF
    rnd = random.Random(7)
    alpha = 'ACGTacgtNn \t\x0b\x0c\x1c\x1f#>xyz-*'
    for case in range(600):
        parts = []
        if rnd.random() < 0.3:
            parts.append(''.join(rnd.choice(alpha) for _ in range(rnd.randint(0, 8))) + '\n')
        for r in range(rnd.randint(0, 6)):
            parts.append('>' + rnd.choice(['', ' ', '\t']) + rnd.choice(['g%d' % rnd.randint(0, 4), 'name x', 'a\tb', 'x']) + rnd.choice(['', ' desc', '  ']) + '\n')
            for _ in range(rnd.randint(0, 4)):
                line = ''.join(rnd.choice(alpha) for _ in range(rnd.randint(0, 12)))
                parts.append(('A' + line if line.startswith('>') else line) + ('\n' if rnd.random() < 0.95 else ''))
                if not parts[-1].endswith('\n'):
                    break
            if not parts[-1].endswith('\n'):
                break
        text = ''.join(parts)
        def outcome(f, x):
            try:
                d = f(x)
                return list(d.items())
            except IndexError:                      # (a header without a name: the reference fails on it the same way)
                return 'no name'
        assert outcome(CF._fasta_records, text.encode()) == outcome(CF._fasta_text_records, text), repr(text)
    # many short records (more than the first guess of the record count), through the file readers, plain and gzipped
    text = ''.join('>%d\nacgt\nNN\n' % (i % 4000) for i in range(5000))
    fa, gz = str(tmp_path / 'a.fa'), str(tmp_path / 'a.fa.gz')
    open(fa, 'w').write(text)
    with gzip.open(gz, 'wt') as f:
        f.write(text)
    want = CF._fasta_text_records(text)
    assert len(want) == 4000 and want['7'] == 'ACGTNN'
    assert CF.readFasta(fa) == want and CF.readFasta(gz) == want and CF.readFastq(fa, with_qual=False)[0] == want
    assert CF.readFastq(fa)[1]['7'] == 'IIII!!' and set(CF.readFasta(fa, headOnly=True).values()) == {''}
    # what the library does not take goes the other way: carriage returns (the text reader's universal newlines), non-ASCII text, a header without a name
    assert CF._fasta_records(b'>a x\r\nAC\r\nGT\r>b\rTT\n') == {'a': 'ACGT', 'b': 'TT'}
    assert CF._fasta_records('>gène\nACGT\n'.encode()) == {'gène': 'ACGT'}
    with pytest.raises(IndexError):
        CF._fasta_records(b'>\nACGT\n')
    with pytest.raises(IndexError):
        CF._fasta_text_records('>\nACGT\n')


def test_argsort_object_order_equals_numpy_on_ties():
    """pep_argsort_object_order against np.argsort of the object column itself: sizes around the sort's insertion-sort threshold, many ties, sorted and
    reversed input, the sizes of a genome's groups (the .tab store's rows of equal score keep the order numpy's sort leaves them in, PEPPAN.py:957-960)"""
    from peppan_amd import _native as N
    rng = np.random.default_rng(0)
    for case in range(400):
        n = int(rng.choice([0, 1, 2, 5, 15, 16, 17, 18, 33, 100, 500, 3000, 6600]))
        kind = case % 5
        if kind == 0:
            x = rng.random(n)
        elif kind == 1:
            x = rng.integers(0, max(1, n // 4 + 1), size=n).astype(np.float64)
        elif kind == 2:
            x = np.round(rng.random(n) * 50) / 7.
        elif kind == 3:
            x = np.sort(rng.integers(0, 10, size=n).astype(np.float64))[::-1].copy()
        else:
            x = np.repeat(rng.random(max(1, n // 3 + 1)), 3)[:n].copy()
            rng.shuffle(x)
        assert np.array_equal(N.argsort_object_order(-x), np.argsort((-x).astype(object))), (case, n, kind)
    assert np.array_equal(N.argsort_object_order(np.array([1., np.nan, 0.])), np.argsort(np.array([1., np.nan, 0.]).astype(object)))      # (NaN: numpy's own way)


def test_overlaps_of_several_tables_in_one_sweep_equal_one_sweep_per_table():
    """mapfilters.overlaps_tables (the genomes of a batched search swept by ONE K11 call) against overlaps_table per table: tables that lost rows to
    the filters (row ids with holes, larger than the table), several contigs per table, an empty table in between, the batch quirk"""
    from oracle import oracle as O
    from peppan_amd import mapfilters
    from peppan_amd.hittable import HitTable
    rng = np.random.default_rng(4)

    def table(n, n_contigs, id_span):
        ri = rng.integers(0, n_contigs, size=n)
        ss = rng.integers(1, 4000, size=n)
        ln = rng.integers(50, 900, size=n)
        rev = rng.random(n) < 0.4
        se = ss + ln
        runs = np.ones(n, dtype=np.int64)
        T = HitTable(list(range(5)), [100 + c for c in range(n_contigs)], rng.integers(0, 5, size=n), ri, np.full(n, 0.9), ln, np.zeros(n, int), np.zeros(n, int),
                     np.ones(n, int), ln, np.where(rev, se, ss), np.where(rev, ss, se), np.zeros(n), ln.astype(float), np.full(n, 1000), np.full(n, 10 ** 6),
                     (ln.astype(np.uint32) << 2), np.arange(n), runs, rid=np.sort(rng.choice(id_span, size=n, replace=False)))
        return T
    tables = [table(200, 3, 500), table(0, 1, 1), table(350, 1, 351), table(60, 6, 4000)]
    for ovl_l, ovl_p, batch in ((300, 0.6, 1000000), (50, 0.1, 1000000), (50, 0.1, 37)):
        want = [mapfilters.overlaps_table(T, ovl_l, ovl_p, batch=batch, sweep=O.overlaps_sweep) for T in tables]
        got = mapfilters.overlaps_tables(tables, ovl_l, ovl_p, O.overlaps_sweep, batch=batch)
        assert len(got) == 4 and sum(map(len, want)) > 100
        for w, g in zip(want, got):
            assert np.array_equal(np.asarray(w).reshape(-1, 3), np.asarray(g).reshape(-1, 3))


def test_similar_classify_equals_its_numpy_statement():
    """pep_similar_classify (host C++, one pass) against pipeline._classify_rows (the numpy form of PEPPAN.py:244-263) over random tables: coordinates
    on both strands, every frame relation, identities on both sides of the threshold, equal and unequal lengths, ranks of all three orders"""
    from peppan_amd import _native as N, pipeline as PL

    class Cols(object):                                               # (the columns the two functions read, as a HitTable holds them)
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def __len__(self):
            return len(self.iden)
    rng = np.random.default_rng(5)
    seen_actions = set()
    for rep in range(20):
        n = int(rng.integers(0, 3000))
        ql, sl = rng.integers(60, 400, size=n), rng.integers(60, 400, size=n)
        same = rng.random(n) < 0.4
        sl[same] = ql[same]
        qs = rng.integers(1, 30, size=n)
        qe = np.minimum(ql, qs + rng.integers(20, 400, size=n))
        a, b = rng.integers(1, 30, size=n), rng.integers(20, 400, size=n)
        lo, hi = np.minimum(a, sl), np.minimum(sl, a + b)
        rev = rng.random(n) < 0.3
        ss, se = np.where(rev, hi, lo), np.where(rev, lo, hi)
        full = rng.random(n) < 0.3                                    # full-length in-frame rows: the absorb branches
        qs[full], qe[full], ss[full], se[full] = 1, ql[full], 1, sl[full]
        iden = np.round(rng.uniform(0.8, 1.0, size=n), 3)
        q, r = rng.integers(0, 50, size=n), rng.integers(0, 50, size=n)
        rank = rng.integers(0, 3, size=50)
        T = Cols(iden=iden, qs=qs, qe=qe, ss=ss, se=se, ql=ql, sl=sl)
        for near, cover in ((0.9, 0.8), (0.95, 0.5), (0.85, 1.0)):
            want = PL._classify_rows(T, rank[q], rank[r], q, r, near, cover)
            got = N.similar_classify(T, q, r, rank[q] >= rank[r], rank[q] <= rank[r], near, cover)
            for w, g in zip(want, got):
                assert w.dtype == g.dtype and np.array_equal(w, g)
            seen_actions.update(want[0].tolist())
    assert seen_actions == {0, 1, 2, 3}                                # every action occurs


def _random_hits(rng, n, n_q, n_t, tool):
    from peppan_amd import _native as N
    h = np.zeros(n, dtype=N.HIT_DTYPE)
    h['q'], h['t'] = rng.integers(0, n_q, n), rng.integers(0, n_t, n)
    h['q_start'] = rng.integers(1, 100, n)
    h['q_end'] = h['q_start'] + rng.integers(5, 300, n)
    h['t_start'] = rng.integers(1, 100, n)
    h['t_end'] = h['t_start'] + rng.integers(5, 300, n)
    h['score'] = rng.integers(20, 2000, n)
    h['aln_len'] = rng.integers(30, 400, n)
    h['nm'] = (h['aln_len'] * rng.random(n) * 0.6).astype(np.uint32)
    h['n_ident'] = h['aln_len'] - h['nm']
    h['cigar_runs'] = rng.integers(1, 6, n)
    h['cigar_off'] = np.concatenate([[0], np.cumsum(h['cigar_runs'])[:-1]])
    cigar = (rng.integers(1, 120, int(h['cigar_runs'].sum())).astype(np.uint32) << 2) | rng.choice([0, 0, 0, 1, 2], int(h['cigar_runs'].sum())).astype(np.uint32)
    return h, cigar


def test_host_chain_passes_do_not_depend_on_their_thread_count():
    """pep_table_from_hits (both tools, rows failing the cuts in every chunk), pep_cols_fix_end and pep_cols_gather over tables beyond the 16 384 rows from which
    they use threads: one thread and four give the same columns, arenas and counts; pep_cols_order against numpy's lexsort (its packed-record radix path, and the
    wide-code path behind it)"""
    from peppan_amd import _native as N
    rng = np.random.default_rng(77)
    n, n_q, n_t = 70001, 3000, 5000
    q_len, r_len = rng.integers(300, 3000, n_q), rng.integers(3000, 9000, n_t)
    results = {}
    before = N.set_host_threads(1)
    try:
        for threads in (1, 4):
            N.set_host_threads(threads)
            out = []
            # translated tool
            rng = np.random.default_rng(5)                                   # (the same inputs for both thread counts)
            h, cigar = _random_hits(rng, n, n_q, n_t, 0)
            qm = np.zeros(n_q, dtype=N.QUERY_META_DTYPE); qm['seq'] = np.arange(n_q); qm['frame'] = 1
            tm = np.zeros(n_t, dtype=N.TARGET_META_DTYPE); tm['seq'] = np.arange(n_t); tm['frame'] = rng.integers(1, 7, n_t); tm['chunk_off'] = rng.integers(0, 50, n_t)
            cols, arena = N.table_from_hits(0, h, cigar, q_len, r_len, 0.55, 60., 0.02, q_meta=qm, t_meta=tm)
            assert 0.2 * n < len(cols['qs']) < 0.95 * n                      # (the cuts bite in every chunk)
            out.append(({k: v.copy() for k, v in cols.items()}, arena.copy()))
            # nucleotide tool
            h, cigar = _random_hits(np.random.default_rng(6), n, n_q, n_t, 1)
            cols, arena = N.table_from_hits(1, h, cigar, q_len, r_len, 0.55, 60., 0.02, t_seq=np.arange(n_t), t_rev=rng.integers(0, 2, n_t).astype(bool), evalue=rng.random(n))
            assert 0.2 * n < len(cols['qs']) < 0.95 * n
            out.append(({k: v.copy() for k, v in cols.items()}, arena.copy()))
            # fixEnd over the nucleotide table (ends near the query's: many rows change), then a gather of every column
            m = len(cols['qs'])
            work = {k: np.ascontiguousarray(v.copy()) for k, v in cols.items()}
            work['qs'] = np.minimum(work['qs'], rng.integers(1, 12, m)); work['ql'] = work['qe'] + rng.integers(0, 12, m)
            work['ss'] = work['ss'] + 20; work['se'] = work['ss'] + (work['qe'] - work['qs']); work['sl'] = work['se'] + rng.integers(0, 12, m)
            fixed_arena = N.cols_fix_end(work, arena, 6., 6.)
            out.append((work, fixed_arena.copy()))
            idx = rng.permutation(m)[:m - 17]
            names = sorted(work)
            out.append((dict(zip(names, N.cols_gather([work[k] for k in names], idx))), np.zeros(0)))
            results[threads] = out
    finally:
        N.set_host_threads(before)
    for (c1, a1), (c4, a4) in zip(results[1], results[4]):
        assert sorted(c1) == sorted(c4)
        for k in c1:
            assert c1[k].dtype == c4[k].dtype and np.array_equal(c1[k], c4[k]), k
        assert np.array_equal(a1, a4)
    # the order: (query code, reference code, score), stable
    for n_rows, q_hi, r_hi in ((70001, 10000, 10000), (50000, 3, 2), (40000, 1 << 20, 1 << 20), (1, 5, 5), (0, 1, 1), (30000, 1, 1)):
        q, r = rng.integers(0, q_hi, n_rows), rng.integers(0, r_hi, n_rows)
        score = rng.integers(0, 40, n_rows).astype(np.float64) / 4
        assert np.array_equal(N.cols_order(q, r, score), np.lexsort((score, r, q))), (n_rows, q_hi, r_hi)


def test_lex_order_equals_numpy_lexsort():
    """pep_lex_order (the sorts in front of -f and -m, uberBlast.py:421, 455): numpy.lexsort's order for int64 keys - ties, constant keys, negative values, a handful of
    rows and 100 000, a key whose range is beyond the radix passes (numpy's sort takes over), a key that is not int64"""
    from peppan_amd import _native as N
    rng = np.random.default_rng(8)
    for n in (0, 1, 63, 64, 1000, 12000, 100000):
        for trial in range(3):
            q, r = rng.integers(0, 10000 if trial else 3, n), rng.integers(0, 3, n)
            ss, qs = rng.integers(-2200000, 2200000, n), rng.integers(1, 1000 if trial < 2 else 2, n)
            assert np.array_equal(N.lex_order((qs, ss, q, r)), np.lexsort((qs, ss, q, r))), n
            assert np.array_equal(N.lex_order((qs, ss, r, q)), np.lexsort((qs, ss, r, q))), n
            wide = rng.integers(-2 ** 50, 2 ** 50, n)
            assert np.array_equal(N.lex_order((qs, wide, q)), np.lexsort((qs, wide, q))), n
            assert np.array_equal(N.lex_order((qs.astype(np.float64), q)), np.lexsort((qs.astype(np.float64), q))), n


def test_crc32_equals_zlib():
    """pep_crc32 (the store members' checksum: carry-less multiplication where the CPU has it, zlib's otherwise): zlib.crc32 for every length around the
    folding steps (16, 64 bytes), for unaligned starts, continued from a running value, and inside pep_pack_member for both coders"""
    import zlib
    from peppan_amd import _native as N
    rng = np.random.default_rng(21)
    blob = bytes(rng.integers(0, 256, 300000).astype(np.uint8))
    for n in list(range(0, 200)) + [255, 256, 257, 1023, 4096, 65535, 65536, 65537, 299999]:
        for start in (0, 1, 7):
            d = blob[start:start + n]
            assert N.crc32(d) == zlib.crc32(d), (n, start)
            cut = len(d) // 3
            assert N.crc32(d[cut:], N.crc32(d[:cut])) == zlib.crc32(d), (n, start)
    for coder in (0, 1):
        for d in (b'', b'abc', blob[:5000], blob[:70000] + blob[:70000], bytes(100000)):
            payload, crc = N.pack_member(d, coder)
            assert crc == zlib.crc32(d) and zlib.decompress(payload, -15) == d
