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
