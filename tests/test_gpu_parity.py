"""GPU parity tests proper: every call goes through the C ABI (libpeppan_hip.so) and is compared
bit-exactly with the CPU oracle on the same seeded inputs."""
import os
import numpy as np
import pytest
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from peppan_amd import _native as N
    c = N.Context(0)
    yield c
    c.close()


def _cmp_hits(gh, gc, oh, oc):
    assert len(gh) == len(oh), (len(gh), len(oh))
    for f in ('q', 't', 'q_start', 'q_end', 't_start', 't_end', 'score', 'nm', 'n_ident', 'aln_len', 'cigar_runs', 'bin', 'cigar_off', 'cells'):
        assert np.array_equal(gh[f], oh[f]), f
    assert np.array_equal(gc, oc)


@pytest.mark.parametrize('use_lds', [1, 0])
@pytest.mark.parametrize('min_id,min_qcov,top_k', [(0., 0., 10), (45., 25., 10), (30., 10., 2)])
def test_search_protein_families(ctx, use_lds, min_id, min_qcov, top_k):
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    prots = synth.make_proteins(240, length=(60, 420), seed=11, family=4, sub=0.25)
    ctx.set_query_aa(prots)
    ctx.set_ref_aa(prots)
    gh, gc, st = ctx.search(N.default_params(min_id, min_qcov, top_k, 5, use_lds=use_lds))
    oh, oc, ost = O.search(prots, prots, O.default_params(min_id, min_qcov, top_k, 5))
    assert st['candidates'] == ost['candidates'] and st['pairs'] == ost['pairs'] and st['cells'] == ost['cells']
    _cmp_hits(gh, gc, oh, oc)
    assert len(gh) > 300


def test_gapless_shortcut_counts_and_parity(ctx):
    """rule 5a: pairs whose alignment is one ungapped run are settled without a traceback sweep - the same pairs in the kernel and in the
    oracle (counted on both sides), and the table stays bit-exact.  Families without indels (most pairs gapless), with indels, the 32-bit
    passes, the nucleotide configuration (hsp_mode 1, long pairs one per wavefront) and sequences that start / end inside the run."""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    for kw, forced32 in ((dict(sub=0.2, indel=0.), False), (dict(sub=0.25), False), (dict(sub=0.2, indel=0.), True)):
        prots = synth.make_proteins(300, length=(40, 500), seed=23, family=5, **kw)
        prots[5] = prots[6][7:]                 # a member that is a suffix / prefix of another: runs that touch row 0 / column 0 / the last cell
        prots[11] = prots[10][:-9]
        ctx.set_query_aa(prots)
        ctx.set_ref_aa(prots)
        p = N.default_params(30., 10., 10, 5)
        p.reserved[1] = 1 if forced32 else 0
        gh, gc, st = ctx.search(p)
        O.trace_counts(True)
        oh, oc, ost = O.search(prots, prots, O.default_params(30., 10., 10, 5))
        tc = O.trace_counts()
        _cmp_hits(gh, gc, oh, oc)
        assert st['tracebacks'] == tc['traced'] and st['tracebacks_gapless'] == tc['gapless'], (st['tracebacks'], st['tracebacks_gapless'], tc)
        assert tc['gapless'] >= 300 and (kw.get('indel', 1) != 0. or tc['gapless'] > 0.8 * tc['traced'])
    names, seqs = synth.make_genes(120, 0, seed=4)
    codes = [O.nt_codes(s.decode()) for s in seqs]
    rc = [(3 - c[::-1]).astype(np.uint8) for c in codes]
    ctx.set_query_aa(codes)
    ctx.set_ref_aa(codes + rc)
    gh, gc, st = ctx.search(N.nucleotide_params(60., 20.))
    O.trace_counts(True)
    oh, oc, ost = O.search(codes, codes + rc, O.params_from(N.nucleotide_params(60., 20.)))
    tc = O.trace_counts()
    _cmp_hits(gh, gc, oh, oc)
    assert st['tracebacks'] == tc['traced'] and st['tracebacks_gapless'] == tc['gapless'] >= 120


def test_search_ragged_and_empty(ctx):
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    prots = synth.make_proteins(40, length=(10, 90), seed=3, family=2, sub=0.1)
    prots[3] = prots[3][:0]                       # empty sequence
    prots[7] = prots[7][:5]                       # shorter than any seed
    prots[9] = np.full(300, 23, np.uint8)         # all X: never seeds
    long_ = synth.make_proteins(2, length=2600, seed=5, family=2, sub=0.1)    # exceeds the LDS staging window -> global path
    qs, ts = prots + long_, long_ + prots[::-1]
    ctx.set_query_aa(qs)
    ctx.set_ref_aa(ts)
    gh, gc, st = ctx.search(N.default_params(0., 0., 10, 5))
    oh, oc, ost = O.search(qs, ts, O.default_params(0., 0., 10, 5))
    _cmp_hits(gh, gc, oh, oc)
    assert len(gh) > 20
    # no queries / no targets
    ctx.set_query_aa([])
    gh, gc, st = ctx.search(N.default_params())
    assert len(gh) == 0 and len(gc) == 0


def test_search_offdiagonal_bands(ctx):
    """a short query inside a long target at various offsets: exercises bands far from the main diagonal,
    both signs, and bin boundaries"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    aa = np.frombuffer(b'ARNDCQEGHILKMFPSTWYV', dtype=np.uint8) - 65
    qs, ts = [], []
    for off in (0, 1, 31, 32, 33, 63, 64, 65, 100, 500, 997):
        q = aa[rng.integers(0, 20, 120)]
        t = np.concatenate([aa[rng.integers(0, 20, off)], q, aa[rng.integers(0, 20, 37)]])
        qs.append(q); ts.append(t.astype(np.uint8))
        qs.append(t.astype(np.uint8)); ts.append(q)       # and the transposed case (negative diagonals)
    ctx.set_query_aa(qs)
    ctx.set_ref_aa(ts)
    gh, gc, st = ctx.search(N.default_params(0., 0., 50, 1))
    oh, oc, ost = O.search(qs, ts, O.default_params(0., 0., 50, 1))
    _cmp_hits(gh, gc, oh, oc)
    assert len(gh) >= 22


def test_k1_translation_golden_and_oracle(ctx):
    from oracle import oracle as O
    g = load_golden('g02_rundiamond.json')
    qn, rn = sorted(g['query']), sorted(g['ref'])
    for frames, key in ((6, '7'), (3, 'F')):
        ctx.set_query_nt([g['query'][n] for n in qn], 11)
        ctx.set_ref_nt([g['ref'][n] for n in rn], frames, 11)
        ctx.translate()
        qm, tm = ctx.query_meta(), ctx.target_meta()
        qa, qo = ctx.query_aa()
        ta, to = ctx.target_aa()
        q_txt = ''.join('>{0}:{1}\n{2}\n'.format(qn[m['seq']], m['frame'], ''.join(chr(65 + c) for c in qa[int(qo[i]):int(qo[i + 1])])) for i, m in enumerate(qm))
        recs = ['>{0}:{1}:{2}\n{3}\n'.format(rn[m['seq']], m['frame'], m['chunk_off'], ''.join(chr(65 + c) for c in ta[int(to[i]):int(to[i + 1])])) for i, m in enumerate(tm)]
        assert q_txt == g['out'][key]['qryAA']
        assert [''.join(recs[i::5]) for i in range(5)] == g['out'][key]['refAA']
    # ambiguous bases, gaps, lower case, table 4, lengths not divisible by 3 (golden G1 inputs) against the oracle restatement
    t = load_golden('g01_transeq.json')
    names = sorted(t['seqs'])
    for table in (11, 4):
        ctx.set_query_nt([t['seqs'][n] for n in names], table)
        ctx.set_ref_nt([t['seqs'][n] for n in names], 6, table)
        ctx.translate()
        qa, qo = ctx.query_aa()
        for i, m in enumerate(ctx.query_meta()):
            f, s = O.query_frame(t['seqs'][names[i]], table)
            assert m['frame'] == f and ''.join(chr(65 + c) for c in qa[int(qo[i]):int(qo[i + 1])]) == s.replace('-', 'X')
        ta, to = ctx.target_aa()
        exp = []
        for n in names:
            for f, aa_ in zip(range(1, 7), O.translate_frames(t['seqs'][n], range(1, 7), table)):
                exp += [(n, f, o, c.replace('-', 'X')) for o, c in O.ref_chunks(aa_)]
        got = [(names[m['seq']], int(m['frame']), int(m['chunk_off']), ''.join(chr(65 + c) for c in ta[int(to[i]):int(to[i + 1])])) for i, m in enumerate(ctx.target_meta())]
        assert got == exp


def test_k1_random_shapes_vs_oracle(ctx):
    """K1 on shapes the byte-parallel translation has to get right: lengths 0 .. 40 and around the 512-codon chunk and the 1000-residue
    cut, unaligned starts, lower case, N and '-' anywhere, the first sequence of the buffer read backwards (frames 4-6), both tables"""
    from oracle import oracle as O
    rng = np.random.default_rng(41)
    alphabet = np.array(list('ACGT' * 12 + 'acgt' * 3 + 'NnRY-'))
    lengths = list(range(0, 41)) + [1533, 1534, 1535, 1536, 1537, 1538, 1539, 1540, 2999, 3000, 3001, 3002, 3003, 3004, 3073, 4611, 6150] + \
        [int(x) for x in rng.integers(41, 5000, 60)]
    seqs = [''.join(rng.choice(alphabet, size=n)).encode() for n in lengths]
    clean = [bytes(rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), size=n)) for n in (7, 1536, 3001, 4000, 6200)]       # stop-free stretches are rare in random text:
    seqs += [s.replace(b'TAA', b'TCA').replace(b'TAG', b'TCG').replace(b'TGA', b'TCA') for s in clean]                   # long frames without a cut
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    for table in (11, 4):
        for frames in (6, 3):
            ctx.set_query_nt(seqs, table)
            ctx.set_ref_nt(seqs, frames, table)
            ctx.translate()
            qa, qo = ctx.query_aa()
            qm = ctx.query_meta()
            assert len(qm) == len(seqs)
            for i, m in enumerate(qm):
                f, aa = O.query_frame(seqs[i].decode(), table)
                assert m['frame'] == f and ''.join(chr(65 + c) for c in qa[int(qo[i]):int(qo[i + 1])]) == aa.replace('-', 'X'), (table, i, len(seqs[i]))
            ta, to = ctx.target_aa()
            exp = []
            for n, sq in enumerate(seqs):
                for f, aa_ in zip(range(1, frames + 1), O.translate_frames(sq.decode(), range(1, frames + 1), table)):
                    exp += [(n, f, o, c.replace('-', 'X')) for o, c in O.ref_chunks(aa_)]
            got = [(int(m['seq']), int(m['frame']), int(m['chunk_off']), ''.join(chr(65 + c) for c in ta[int(to[i]):int(to[i + 1])])) for i, m in enumerate(ctx.target_meta())]
            assert got == exp, (table, frames)


def test_k1_chunks_of_long_contigs_vs_oracle(ctx):
    """K1's chunk lists for LONG reference sequences (round 6: frames beyond 98 304 characters are cut into segments whose speculative chunk chains are joined to the
    true chain, k1_ref_chunks_spec / _join) against the reference's regex over the whole frame (oracle.ref_chunks): contigs of random text (a stop every ~21 codons),
    of gene-like text (open reading frames of hundreds of codons in one frame or the other), with a stop-free stretch longer than a segment, with runs of N, of lengths
    around the segment size and its multiples; the residues too"""
    from oracle import oracle as O
    from peppan_amd import synth
    rng = np.random.default_rng(97)
    acgt = np.frombuffer(b'ACGT', dtype=np.uint8)
    names, genes = synth.make_genes(700, 0, seed=3)
    def gene_like(n_nt):                                  # genes on both strands with short spacers: long ORFs in every frame now and then
        parts, total = [], 0
        while total < n_nt:
            g = np.frombuffer(genes[int(rng.integers(0, len(genes)))], dtype=np.uint8)
            if rng.random() < 0.5:
                g = np.array([84, 0, 71, 0, 0, 0, 67, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 65], dtype=np.uint8)[g[::-1] - 65]       # reverse complement of ACGT
            sp = rng.choice(acgt, size=int(rng.integers(20, 200)))
            parts += [g, sp]
            total += len(g) + len(sp)
        return np.concatenate(parts)[:n_nt].tobytes()
    K = 49152 * 3
    no_stop = bytes(rng.choice(acgt, size=3 * 60000)).replace(b'TAA', b'TCA').replace(b'TAG', b'TCG').replace(b'TGA', b'TCA')      # (stop-free in frame 1 only where no codon boundary was joined: good enough)
    seqs = [bytes(rng.choice(acgt, size=n)) for n in (2 * K - 5, 2 * K, 2 * K + 1, 2 * K + 2, 2 * K + 3, 2 * K + 7, 3 * K + 11, 5 * K - 1)]
    seqs += [gene_like(n) for n in (2 * K + 100, 700001, 1234567)]
    seqs += [gene_like(400000) + no_stop + gene_like(250000), bytes(rng.choice(acgt, size=350000)) + b'N' * 5000 + bytes(rng.choice(acgt, size=300000))]
    seqs += [genes[0], genes[1], gene_like(90000)]        # short ones beside them (the plain walk)
    for frames in (6, 3):
        ctx.set_ref_nt(seqs, frames, 11)
        ctx.set_query_nt(genes[:3], 11)
        ctx.translate(force=True)
        ta, to = ctx.target_aa()
        tm = ctx.target_meta()
        exp = []
        for n, sq in enumerate(seqs):
            for f, aa_ in zip(range(1, frames + 1), O.translate_frames(sq.decode(), range(1, frames + 1), 11)):
                exp += [(n, f, o, len(c)) for o, c in O.ref_chunks(aa_)]
        got = [(int(m['seq']), int(m['frame']), int(m['chunk_off']), int(to[i + 1]) - int(to[i])) for i, m in enumerate(tm)]
        assert len(got) == len(exp) and got == exp, frames
        # the residues of a sample of chunks
        pick = rng.choice(len(got), size=200, replace=False)
        frames_aa = {}
        for i in pick.tolist():
            n, f, o, ln = got[i]
            if (n, f) not in frames_aa:
                frames_aa[(n, f)] = O.translate_frames(seqs[n].decode(), [f], 11)[0].replace('-', 'X')
            assert ''.join(chr(65 + c) for c in ta[int(to[i]):int(to[i + 1])]) == frames_aa[(n, f)][o:o + ln], (frames, i)
    assert max(len(s) for s in seqs) > 1200000 and sum(1 for n, f, o, ln in got if ln > 5000) >= 1


def test_search_from_nucleotides_1k(ctx):
    """BASELINE config 1k synthetic 1 kb genes, all-vs-all, from nucleotides: K1..K8 against the oracle"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    names, seqs = synth.make_genes(1000, 1002, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])      # the reference writes FASTA in sorted(name) order
    nts = [seqs[i] for i in order]
    ctx.set_query_nt(nts, 11)
    ctx.set_ref_nt(nts, 6, 11)
    gh, gc, st = ctx.search(N.default_params(45., 25., 10, 5))
    q_aa = [O.aa_codes(O.query_frame(s.decode(), 11)[1].replace('-', 'X')) for s in nts]
    t_aa = []
    for s in nts:
        for aa_ in O.translate_frames(s.decode(), range(1, 7), 11):
            t_aa += [O.aa_codes(c.replace('-', 'X')) for o, c in O.ref_chunks(aa_)]
    oh, oc, ost = O.search(q_aa, t_aa, O.default_params(45., 25., 10, 5))
    _cmp_hits(gh, gc, oh, oc)
    assert len(gh) > 2000 and st['cells'] == ost['cells']


def test_k7_rescore_counts(ctx):
    from peppan_amd import _native as N
    from oracle import oracle as O
    g = load_golden('g05_rescore.json')
    qn, rn = sorted(g['query']), sorted(g['ref'])
    qi, ri = {n: i for i, n in enumerate(qn)}, {n: i for i, n in enumerate(rn)}
    ctx.set_query_nt([g['query'][n] for n in qn], 11)
    ctx.set_ref_nt([g['ref'][n] for n in rn], 6, 11)
    opc = {'M': 0, 'I': 1, 'D': 2}
    hits = np.zeros(len(g['table']), dtype=N.NT_HIT_DTYPE)
    cig = []
    for k, row in enumerate(g['table']):
        hits[k] = (qi[row[0]], ri[row[1]], row[6], row[7], row[8], row[9], len(row[14]), 0, len(cig))
        cig += [(n << 2) | opc[o] for n, o in row[14]]
    out = ctx.rescore_nt(hits, np.array(cig, dtype=np.uint32))
    for k, row in enumerate(g['table']):
        exp = O.rescore_counts(O.nt_encode_rescore(g['query'][row[0]].upper()), O.nt_encode_rescore(g['ref'][row[1]].upper()), row[6], row[8], row[9],
                               np.array(cig[hits[k]['cigar_off']:hits[k]['cigar_off'] + len(row[14])], dtype=np.uint32))
        assert np.array_equal(out[k], exp), row[:2]
    # a CIGAR that does not fit its coordinates is an error, not a wild read
    bad = hits[:1].copy(); bad['re'] += 7
    with pytest.raises(N.PepError):
        ctx.rescore_nt(bad, np.array(cig, dtype=np.uint32))


def test_k10_components(ctx):
    from oracle import oracle as O
    rng = np.random.default_rng(10)
    for n, m in ((1, 0), (60, 40), (5000, 3000), (200000, 350000)):
        a, b = rng.integers(0, n, m), rng.integers(0, n, m)
        assert np.array_equal(ctx.components(n, a, b), O.components(n, a, b))
    g = load_golden('g11_groups.json')
    for case in g['cases']:
        clu, bsn = np.array(case['clu']), np.array(case['bsn'])
        e = np.vstack([clu[:, :2], bsn[bsn[:, 2] > 0][:, :2]])
        lab = ctx.components(60, e[:, 0], e[:, 1])
        got = {}
        for i, l in enumerate(lab):
            got.setdefault(int(l), set()).add(i)
        assert {frozenset(s) for s in got.values() if len(s) > 1} == {frozenset(m) for _, m in case['groups']}


def _write_fasta(path, names, seqs):
    with open(path, 'w') as f:
        for n, s in zip(names, seqs):
            f.write('>%s\n%s\n' % (n, s.decode()))


def test_uberblast_dropin_self_search(tmp_path, monkeypatch):
    """the call PEPPAN makes for the exemplar all-vs-all (PEPPAN.py:229-230, minus --blastn): whole 16-column
    table from the HIP path == the same host code over the oracle, and the union-find partitions agree"""
    import io, contextlib
    from peppan_amd import uberBlast as UB, synth
    from oracle_context import OracleContext
    monkeypatch.chdir(tmp_path)
    names, seqs = synth.make_genes(240, 0, seed=77)           # log-normal lengths
    names = [str(1000 + 7 * i) for i in range(len(seqs))]     # string order != numeric order
    fa = str(tmp_path / 'ex.fa')
    _write_fasta(fa, names, seqs)
    argv = '-r {0} -q {0} --diamond -s 1 --min_id 0.45 --min_cov 50 -t 2 --min_ratio 0.25 -e 3,3 -p --gtable 11'.format(fa).split()
    with contextlib.redirect_stderr(io.StringIO()):
        gpu = UB.uberBlast(argv)
        octx = OracleContext()
        monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
        ora = UB.uberBlast(argv)
    assert gpu.shape == ora.shape and gpu.shape[0] > 300 and gpu.shape[1] == 16
    for a, b in zip(gpu.tolist(), ora.tolist()):
        assert a == b
    # floats within 1e-4 is the stated tolerance; they are in fact identical (integer counts + float64 on the host)
    assert all(isinstance(r[14], str) for r in gpu.tolist())


def test_uberblast_dropin_genome_mapping(tmp_path, monkeypatch):
    """genes against a contig carrying them on both strands: reverse frames, chunk offsets, -f -m -O post-filters"""
    import io, contextlib
    from peppan_amd import uberBlast as UB, synth, configure
    from oracle_context import OracleContext
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(5)
    names, seqs = synth.make_genes(40, 0, seed=9, family=2)
    spacer = lambda: bytes(rng.choice(list(b'ACGT'), int(rng.integers(50, 300))).tolist())
    contig = spacer()
    for k, s in enumerate(seqs[::2]):
        contig += (s if k % 2 else configure.rc(s.decode()).encode()) + spacer()
    _write_fasta(str(tmp_path / 'genes.fa'), names, seqs)
    _write_fasta(str(tmp_path / 'genome.fa'), ['7:contig1'], [contig])
    argv = ('-r {0} -q {1} -f -m -O --diamond --min_id 0.4 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11'
            .format(tmp_path / 'genome.fa', tmp_path / 'genes.fa')).split()
    with contextlib.redirect_stderr(io.StringIO()):
        gpu_tab, gpu_ovl = UB.uberBlast(argv)
        octx = OracleContext()
        monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
        ora_tab, ora_ovl = UB.uberBlast(argv)
    assert gpu_tab.shape[0] >= 20 and gpu_tab.tolist() == ora_tab.tolist() and gpu_ovl.tolist() == ora_ovl.tolist()
    assert any(r[8] > r[9] for r in gpu_tab.tolist()) and any(r[8] < r[9] for r in gpu_tab.tolist())


@pytest.mark.parametrize('min_id', [1.0, 0.97, 0.9])
def test_k9_linclust_vs_oracle(ctx, min_id):
    from peppan_amd import synth
    from oracle import oracle as O
    names, seqs = synth.make_genes(3000, 0, seed=21)
    seqs = list(seqs) + [seqs[5], seqs[5][:-30], b'ACGT' * 3, b'', b'ACGTNNNNACGTACGTACGTACGTNACGT' * 4, seqs[7][10:]]     # duplicates, fragments, short, empty, ambiguous
    codes = [O.nt_codes(s) for s in seqs]
    g_rep, g_st = ctx.linclust(codes, min_id, 0.9)
    o_rep, o_st = O.linclust(codes, min_id, 0.9)
    assert g_st == o_st
    assert np.array_equal(g_rep, o_rep)
    assert (g_rep[g_rep] == g_rep).all()                      # representatives represent themselves
    if min_id < 1.0:
        assert len(set(g_rep.tolist())) < len(seqs) - 20
    # protein alphabet (clust -a): base 20, k 7
    prots = synth.make_proteins(300, length=(50, 300), seed=4, family=3, sub=0.03)
    aa20 = np.full(26, 20, dtype=np.uint8)
    for i, c in enumerate('ACDEFGHIKLMNPQRSTVWY'):
        aa20[ord(c) - 65] = i
    pc = [aa20[p] for p in prots]
    g_rep, g_st = ctx.linclust(pc, 0.9, 0.8, base=20, k=7, m=20)
    o_rep, o_st = O.linclust(pc, 0.9, 0.8, base=20, k=7, m=20)
    assert np.array_equal(g_rep, o_rep) and g_st == o_st


@pytest.mark.parametrize('translate', [False, True])
def test_getclust_and_iterclust_on_gpu(tmp_path, monkeypatch, translate):
    """clust.getClust / pipeline.iterClust with the GPU clusterer == the same host code over the oracle's relation; translate = clust -a:
    the first-frame proteins are clustered (base 20, k 7; lower-case and comment lines in the file on the nucleotide side)"""
    import io, contextlib
    from peppan_amd import clust as CL, pipeline as PL, linclust as LC, synth
    from oracle import oracle as O
    monkeypatch.chdir(tmp_path)
    names, seqs = synth.make_genes(1200, 0, seed=31)
    names = [str(i) for i in range(len(seqs))]
    with open('genes.fa', 'w') as f:
        f.write('text in front of the first record\n')
        for k, (n, s) in enumerate(zip(names, seqs)):
            t = s.decode()
            if k % 5 == 0:
                t = t.lower()
            lines = [t[j:j + 70] for j in range(0, len(t), 70)]
            if k % 11 == 0:
                lines.insert(1, '# a remark')
            f.write('>%s description\n%s\n' % (n, '\n'.join(lines)))
    aa20 = np.full(256, 20, dtype=np.uint8)
    for i, c in enumerate('ACDEFGHIKLMNPQRSTVWY'):
        aa20[ord(c)] = i

    def oracle_fn(fasta, identity, coverage, n_thread):
        recs = CL.readFasta(fasta)
        if translate:
            rep, _ = O.linclust([aa20[np.frombuffer(s.encode(), dtype=np.uint8)] for _, s in recs], float(identity), float(coverage), base=20, k=7, m=20)
        else:
            rep, _ = O.linclust([O.nt_codes(s) for _, s in recs], float(identity), float(coverage))
        return [(recs[r][0], recs[i][0]) for i, r in enumerate(rep.tolist())]
    out = {}
    for tag, fn in (('gpu', None), ('ora', oracle_fn)):
        with contextlib.redirect_stderr(io.StringIO()):
            groups = [[0, 999999, 10000]]
            ex = PL.iterClust(tag, 'genes.fa', groups, dict(identity=0.9, coverage=0.8, n_thread=1, translate=translate, cluster_fn=fn))
        out[tag] = (open(ex).read(), open(tag + '.clust.tab').read(), np.load(tag + '.clust.npy').tolist())
    assert out['gpu'] == out['ora']
    assert out['gpu'][0].count('>') < 1200 and len(out['gpu'][2]) > 50


def test_nucleotide_search_engine_vs_oracle(ctx):
    """the blastn-like configuration of the engine (base-4 exact 17-mers, +2/-3, gap 6+2k), both strands"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    names, seqs = synth.make_genes(400, 0, seed=41)
    codes = [O.nt_codes(s) for s in seqs]
    rc = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    targets = codes + [rc[c[::-1]] for c in codes]
    p = N.nucleotide_params(70., 25.)
    ctx.set_query_aa(codes)
    ctx.set_ref_aa(targets)
    gh, gc, st = ctx.search(p)
    ms = np.array([O.min_score(len(c), p.dbsize, p.max_evalue, p.ka_lambda, p.ka_k) for c in codes], dtype=np.int32)
    oh, oc, ost = O.search(codes, targets, O.params_from(p), min_scores=ms)
    _cmp_hits(gh, gc, oh, oc)
    assert st['candidates'] == ost['candidates'] and len(gh) > 500
    assert (gh['q'] == gh['t']).sum() == 400           # every gene finds itself on the forward strand


def test_nucleotide_stride_lookup_equals_plain_matcher_and_oracle(ctx):
    """seed_match_stride (14-mer look-ups at every fourth target position + flank verification, round 6) against the plain 17-mer matcher
    (params.reserved[0] = 8) and the oracle: same raw hit COUNT, same target-seed count, same candidates, same table and CIGARs.  Inputs made to
    sit on the rule's edges: exact runs of 14 .. 21 bases planted at every phase of the stride and across the 1 024-position tile borders,
    ambiguous bases inside and next to runs, genes shorter than a word, repeats, both strands, a target set that ends inside a tile"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    rc = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    for rep in range(4):
        names, seqs = synth.make_genes(120 + 40 * rep, 0, seed=90 + rep)
        codes = [O.nt_codes(s) for s in seqs]
        # short genes, genes with ambiguous bases
        codes += [rng.integers(0, 4, size=n).astype(np.uint8) for n in (5, 13, 14, 16, 17, 18, 21, 33)]
        amb = [c.copy() for c in codes[:30]]
        for c in amb:
            c[rng.integers(0, len(c), size=max(1, len(c) // 40))] = 4
        # targets: the genes, pieces of them of every length around the word size planted in random sequence at every offset, both strands
        planted = []
        for ln in range(13, 23):
            for ph in range(4):
                g = codes[int(rng.integers(0, 100))]
                st = int(rng.integers(0, len(g) - ln))
                left = rng.integers(0, 4, size=40 + ph).astype(np.uint8)
                right = rng.integers(0, 4, size=37).astype(np.uint8)
                piece = g[st:st + ln].copy()
                # the bases right outside the run differ from the gene's, so the exact run is ln long (unless the gene goes on by chance further out)
                if st > 0:
                    left[-1] = (g[st - 1] + 1) % 4
                if st + ln < len(g):
                    right[0] = (g[st + ln] + 1) % 4
                planted.append(np.concatenate([left, piece, right]))
        filler = [rng.integers(0, 4, size=int(n)).astype(np.uint8) for n in rng.integers(900, 1200, size=6)]      # moves what follows across tile borders
        targets = codes[:60] + planted + filler + amb + codes[60:]
        targets = targets + [rc[c[::-1]] for c in targets]
        queries = codes + amb
        ctx.set_query_aa(queries)
        ctx.set_ref_aa(targets)
        out = {}
        for flag in (0, 8):
            p = N.nucleotide_params(70., 25.)
            p.reserved[0] = flag
            gh, gc, st = ctx.search(p)
            out[flag] = (gh.tobytes(), gc.tobytes(), st['target_seeds'], st['seed_hits'], st['candidates'], st['pairs'])
        assert out[0] == out[8], (rep, out[0][2:], out[8][2:])
        p = N.nucleotide_params(70., 25.)
        ms = np.array([O.min_score(len(c), p.dbsize, p.max_evalue, p.ka_lambda, p.ka_k) for c in queries], dtype=np.int32)
        oh, oc, ost = O.search(queries, targets, O.params_from(p), min_scores=ms)
        _cmp_hits(np.frombuffer(out[0][0], dtype=N.HIT_DTYPE), np.frombuffer(out[0][1], dtype=np.uint32), oh, oc)
        assert out[0][4] == ost['candidates'] and out[0][3] > 20000
    # the tool's own layout from nucleotide sets (pep_use_nt_as_residues: forward strands, then reverse complements) - here the forward strand of reference gene g IS query g
    # and the matcher counts but drops the gene's diagonal-0 hits against itself, judging the bases next to a look-up word by the target's side alone (self_prepare has settled
    # the candidate): the same table, raw hit count and candidates as the plain matcher and the oracle; also with a reference that differs in a gene, a base, its order and size
    names, seqs = synth.make_genes(260, 0, seed=77)
    seqs = [bytes(x) for x in seqs]
    seqs[5] = seqs[5][:200] + b'NNNN' + seqs[5][204:]
    seqs[6] = seqs[7]
    seqs[8] = b'ACGTACGTACGTACGTAC'
    seqs[9] = b''
    other = list(seqs); other[20] = seqs[21]
    base = list(seqs); base[30] = seqs[30][:40] + (b'A' if seqs[30][40:41] != b'A' else b'C') + seqs[30][41:]
    longer = list(seqs); longer[40] = seqs[40] + b'ACGTTGCA'
    for ref in (seqs, other, base, longer, seqs[::-1], seqs[:-3], seqs + [seqs[0]]):
        res = {}
        for flag in (0, 8):
            p = N.nucleotide_params(70., 25.)
            p.reserved[0] = flag
            ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(ref, 6, 11); ctx.use_nt_as_residues(2)
            gh, gc, st = ctx.search(p)
            res[flag] = (gh.tobytes(), gc.tobytes(), st['target_seeds'], st['seed_hits'], st['candidates'], st['pairs'])
        assert res[0] == res[8], (len(ref), res[0][2:], res[8][2:])
    codes = [O.nt_codes(x.upper()) for x in seqs]
    p = N.nucleotide_params(70., 25.)
    ctx.set_query_nt(seqs, 11); ctx.set_ref_nt(seqs, 6, 11); ctx.use_nt_as_residues(2)
    gh, gc, st = ctx.search(p)
    ms = np.array([O.min_score(len(c), p.dbsize, p.max_evalue, p.ka_lambda, p.ka_k) for c in codes], dtype=np.int32)
    oh, oc, ost = O.search(codes, codes + [rc[c[::-1]] for c in codes], O.params_from(p), min_scores=ms)
    _cmp_hits(gh, gc, oh, oc)
    assert st['candidates'] == ost['candidates'] and (gh['q'] == gh['t']).sum() >= 255


def test_self_search_drops_diagonal_zero_self_hits_exactly(ctx):
    """PEPPAN's hot call searches a gene set against itself (PEPPAN.py:229-230): frame 1 of reference gene g IS query g, and more than half of the raw seed
    hits are a gene against itself on diagonal 0.  Round 6: the matcher counts and drops them (pep_self_map decides on the device which targets repeat a
    query) and self_candidates settles their one candidate from the sequences.  The table, the CIGARs, the candidate count and every seed statistic equal
    the plain stream's (params.reserved[0] = 10) and the oracle's.  The gene set holds what decides gene by gene: genes whose query frame is not 1 (stops in
    frame 1), genes beyond 1 000 codons (several chunks per frame), an empty and a tiny gene, lower-case and ambiguous bases, duplicates, a gene of X runs
    whose self hits do not all pass the pre-filter; then reference sets that differ from the queries in one gene, in one base, in their order and number."""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    names, seqs = synth.make_genes(160, 0, seed=12)
    seqs = [bytes(s) for s in seqs]
    seqs[3] = b'ATGTAATAGTGA' + seqs[3][12:]                         # stops at the start of frame 1: another frame wins or ties
    seqs[4] = seqs[4][:100] + b'TAA' + seqs[4][103:200] + b'TAG' + seqs[4][203:]
    seqs[5] = b'C' + seqs[5]                                          # frame 2 is the reading frame
    seqs[6] = b''
    seqs[7] = b'ATGAAATAA'
    seqs[8] = seqs[8].lower()
    seqs[9] = seqs[9][:50] + b'NNNRY' + seqs[9][55:]
    seqs[10] = seqs[11]                                               # duplicates
    long_codons = rng.integers(0, 61, size=1400)
    sense = [a + b + c for a in 'ACGT' for b in 'ACGT' for c in 'ACGT' if a + b + c not in ('TAA', 'TAG', 'TGA')]
    seqs[12] = ('ATG' + ''.join(sense[i] for i in long_codons) + 'TAA').encode()      # 1 402 codons: frame 1 is cut into chunks
    seqs[13] = seqs[12][:2000] + b'A' + seqs[12][2001:]
    # short stretches of seeding residues between ambiguous codons: self hits whose extension is cut on both sides
    seqs[14] = b'ATG' + b''.join((b'GGTGCTTCTGCAGGTTCAGCTGGT' + b'NNN' * 3) for _ in range(12)) + b'TAA'
    seqs[15] = b'ATG' + b'GCTGGTTCTGCAGGTGCTTCAGGT' * 2 + b'NNN' * 40 + b'TAA'
    p_on, p_off = N.default_params(45., 25., 10, 5), N.default_params(45., 25., 10, 5)
    p_off.reserved[0] = 10

    def oracle_table(q, r):
        q_aa = [O.aa_codes(O.query_frame(s.decode(), 11)[1]) for s in q]
        t_aa = []
        for s in r:
            for aa in O.translate_frames(s.decode(), range(1, 7), 11):
                t_aa += [O.aa_codes(c) for o, c in O.ref_chunks(aa)]
        return O.search(q_aa, t_aa, O.default_params(45., 25., 10, 5))

    def both(q, r, inside=False):
        out = []
        for p in (p_on, p_off):
            ctx.set_query_nt(q, 11)
            ctx.set_ref_nt(r, 6, 11)
            if inside:
                ctx.invalidate_translation()                          # K1 inside the search: the map is made behind it on the stream
            out.append(ctx.search(p))
        (h1, c1, s1), (h0, c0, s0) = out
        _cmp_hits(h1, c1, h0, c0)
        for k in ('query_seeds', 'target_seeds', 'seed_hits', 'candidates', 'pairs', 'tracebacks', 'hits', 'cells'):
            assert s1[k] == s0[k], k
        return h1, c1, s1

    h, c, st = both(seqs, seqs)
    oh, oc, ost = oracle_table(seqs, seqs)
    _cmp_hits(h, c, oh, oc)
    assert st['candidates'] == ost['candidates'] and len(h) > 300
    h2, c2, st2 = both(seqs, seqs, inside=True)
    _cmp_hits(h2, c2, h, c)
    # a second search on the same sets without a new translation
    h3, c3, st3 = ctx.search(p_on)
    _cmp_hits(h3, c3, h, c)
    # reference sets that are not the query set: one gene replaced, one base changed, another order, one gene less, one more; queries that are a part
    other = list(seqs); other[20] = seqs[21]
    base = list(seqs); base[30] = seqs[30][:40] + (b'A' if seqs[30][40:41] != b'A' else b'C') + seqs[30][41:]
    for q, r in ((seqs, other), (seqs, base), (seqs, seqs[::-1]), (seqs, seqs[:-1]), (seqs, seqs + [seqs[0]]), (seqs[:50], seqs), (seqs[40:90], seqs)):
        both(q, r)
    oh, oc, ost = oracle_table(seqs, base)
    h, c, st = both(seqs, base)
    _cmp_hits(h, c, oh, oc)
    assert st['candidates'] == ost['candidates']
    # bigger sets with families, log-normal lengths
    names, big = synth.make_genes(1500, 0, seed=31)
    h, c, st = both(big, big, inside=True)
    assert len(h) > 3000


def test_uberblast_dropin_blastn_and_diamond(tmp_path, monkeypatch):
    """the reference's actual exemplar call: --blastn --diamond -s 1 (PEPPAN.py:229-230), both tools on the GPU"""
    import io, contextlib
    from peppan_amd import uberBlast as UB, synth
    from oracle_context import OracleContext
    monkeypatch.chdir(tmp_path)
    names, seqs = synth.make_genes(160, 0, seed=55)
    names = [str(3 * i + 1) for i in range(len(seqs))]
    fa = str(tmp_path / 'ex.fa')
    _write_fasta(fa, names, seqs)
    argv = '-r {0} -q {0} --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 2 --min_ratio 0.25 -e 3,3 -p --gtable 11'.format(fa).split()
    with contextlib.redirect_stderr(io.StringIO()):
        gpu = UB.uberBlast(argv)
        octx = OracleContext()
        monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
        ora = UB.uberBlast(argv)
    assert gpu.shape == ora.shape and gpu.tolist() == ora.tolist()
    n_self = sum(1 for r in gpu.tolist() if r[0] == r[1])
    assert n_self >= 2 * 160 - 5            # blastn and diamond rows of the same pair coexist (uberBlast.py:343-346, 353)


def test_allgather_from_device_memory_over_rccl():
    """the exchange step with the table left on the GPU (Context.search_on_device -> dist.allgather_hits(on_device=...)): payload assembled
    in device memory, RCCL all-gather (a one-rank group is all a one-GPU box offers: every copy, launch and synchronisation except the
    xGMI transfer), result == the host-staged exchange == the table itself with re-based indices; then K10 and the lazy host copy.
    In a process of its own: torch has to be imported before the library is loaded (its wheel carries its own HIP runtime)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'exchange_device_check.py')], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')))
    assert r.returncode == 0 and 'exchange from device memory: ok' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_public_tools_keep_the_reference_plugin_contract_on_gpu(tmp_path, monkeypatch):
    """the product's PUBLIC tools - blastn + diamond together - return what the `tools` dictionary of uberBlast.py:327 promises
    (tests/plugin_contract.py), and stacked, numbered and sent through the public reScore / fixEnd / returnOverlap they give the table
    RunBlast.run gives, on the GPU"""
    import io, contextlib
    from peppan_amd import uberBlast as UB, synth
    from plugin_contract import tools_then_methods
    monkeypatch.chdir(tmp_path)
    names, seqs = synth.make_genes(200, 0, seed=91)
    names = [str(5 * i + 2) for i in range(len(seqs))]
    fa = str(tmp_path / 'ex.fa')
    _write_fasta(fa, names, seqs)
    with contextlib.redirect_stderr(io.StringIO()):
        via_loop = tools_then_methods(UB.RunBlast(), fa, fa, ['blastn', 'diamond'], 0.45, 50., 0.25, re_score=1, fix_end=(3., 3.))
        via_run = UB.RunBlast().run(fa, fa, ['blastn', 'diamond'], 0.45, 50., 0.25, re_score=1, return_overlap=[False, 300, 0.6], fix_end=[3., 3.])
        a, a_ovl = tools_then_methods(UB.RunBlast(), fa, fa, ['diamondSELF'], 0.45, 50., 0.25, fix_end=(0., 3.), overlap=(300, 0.6))
        b, b_ovl = UB.RunBlast().run(fa, fa, ['diamondSELF'], 0.45, 50., 0.25, return_overlap=[True, 300, 0.6], fix_end=[0., 3.])
    assert via_loop.shape[0] > 400 and via_loop.shape[1] == 16 and via_loop.tolist() == via_run.tolist()
    assert a.tolist() == b.tolist() and a_ovl.tolist() == b_ovl.tolist() and len(a) >= 200


def _random_support_groups(rng, n_groups, long_group=False):
    """groups of forward alignments of one gene pair each, as get_similar meets them: overlapping rows, all three frames, gaps, short and
    long genes; identities with three decimals"""
    from peppan_amd import _native as N
    rows, cigar, off, gq, gr = [], [], [0], [], []
    for g in range(n_groups):
        ql = int(rng.choice([90, 300, 1002, 2400, 9000])) + int(rng.integers(0, 30))
        if long_group and g == 0:
            ql = 120000                                          # more than 32 768 covered positions: the serial leaf path of the summation
        rl = max(30, int(ql * rng.choice([1., 1., 0.9, 1.3, 0.04])))
        n_rows = int(rng.choice([1, 1, 2, 2, 3, 6])) if g % 37 else 49
        for _ in range(n_rows):
            qs = int(rng.integers(1, max(2, ql // 3)))
            rs = qs + int(rng.choice([0, 0, 0, 3, 1, 2, 30]))
            runs, qpos = [], qs
            while qpos < ql - 12 and len(runs) < 40:
                m = int(min(ql - qpos, rng.integers(3, max(4, ql // 2))))
                runs.append((m << 2) | 0)
                qpos += m
                if rng.random() < 0.7 and qpos < ql - 12:
                    gl = int(rng.integers(1, 8))
                    op = int(rng.integers(1, 3))
                    runs.append((gl << 2) | op)
                    if op == 1:
                        qpos += gl
                if rng.random() < 0.15:
                    break
            rows.append((qs, rs, len(runs), 0, len(cigar), round(float(rng.uniform(0.45, 1.0)), 3)))
            cigar += runs
        off.append(len(rows)); gq.append(ql); gr.append(rl)
    return (np.array(rows, dtype=N.SUPPORT_ROW_DTYPE), np.array(cigar, dtype=np.uint32), np.array(off, dtype=np.uint64),
            np.array(gq, dtype=np.uint32), np.array(gr, dtype=np.uint32))


def test_pair_support_kernel_equals_get_similar(ctx):
    """K14 (pep_pair_support) == the reference's get_similar restated with its own dictionary and np.mean (oracle.pair_support,
    PEPPAN.py:195-224): every group value, including the last bit of numpy's pairwise mean before int() truncates it"""
    from peppan_amd import _native as N
    from oracle_context import OracleContext
    rng = np.random.default_rng(14)
    base = dict(match_identity=0.5, incompleteCDS='', match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
    seen = set()
    for k, params in enumerate((base, dict(base, incompleteCDS='sife'), dict(base, match_identity=0.8, match_len1=30., match_prop2=0.1))):
        rows, cigar, off, gq, gr = _random_support_groups(rng, 700, long_group=(k == 0))
        lim = N.support_limits(params)
        got = ctx.pair_support(rows, cigar, off, gq, gr, lim)
        want = OracleContext().pair_support(rows, cigar, off, gq, gr, lim)
        assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
        seen |= {'none' if v == N.SUPPORT_NONE else ('zero' if v == 0 else 'mean') for v in got.tolist()}
        assert (got > 0).sum() > 100
    assert seen == {'none', 'zero', 'mean'}
    # nothing to judge / empty groups
    e = ctx.pair_support(rows[:0], cigar[:0], np.zeros(3, np.uint64), gq[:2], gr[:2], lim)
    assert e.tolist() == [N.SUPPORT_NONE, N.SUPPORT_NONE]
    with pytest.raises(N.PepError):
        ctx.pair_support(np.zeros(300, N.SUPPORT_ROW_DTYPE), cigar, np.array([0, 300], np.uint64), gq[:1], gr[:1], lim)      # > 255 rows in one group


def test_get_similar_pairs_golden_on_gpu(tmp_path, monkeypatch):
    """golden G10 (the reference's own get_similar_pairs on a canned table: returned pairs, rewritten exemplar file, clust.npy) through the
    product path proper: numeric table, host scan, K14 on the GPU, resolve"""
    import copy
    from peppan_amd import pipeline as PL
    g = load_golden('g10_pairs.json')
    for params_in, exp in ((g['params'], g), (g['variant_sife']['params'], g['variant_sife'])):
        cl = tmp_path / 'p.clust.exemplar'
        cl.write_text(g['exemplar_in'])
        np.save(str(tmp_path / 'p.clust.npy'), np.array(g['clust_npy_in'], dtype=int))
        tab = np.empty([len(g['table']), 16], dtype=object)
        for i, r in enumerate(g['table']):
            for j, v in enumerate(r):
                tab[i, j] = v
        monkeypatch.setattr(PL, 'uberBlast', lambda argv, pool=None, as_table=False: copy.deepcopy(tab))
        res = PL.get_similar_pairs(str(cl), {int(k): v for k, v in g['priorities'].items()}, dict(params_in, clust=str(cl)))
        assert res.tolist() == exp['pairs'] and cl.read_text() == exp['exemplar_out']
        assert np.load(str(tmp_path / 'p.clust.npy'), allow_pickle=True).tolist() == exp['clust_npy_out']


@pytest.mark.parametrize('nucl', [False, True])
def test_config1_1k_genes_cluster_membership_bit_exact(tmp_path, monkeypatch, nucl):
    """BASELINE configs[1]: 1k synthetic 1 kb genes, all-vs-all on one MI355X, cluster membership bit-exact vs the CPU path.
    Whole hot path through the reference-shaped entry points: iterClust -> get_similar_pairs -> get_gene_group.
    nucl: PEPPAN's --nucl mode (PEPPAN.py:225-227: the nucleotide tool alone, no diamond, no rescoring)."""
    import io, contextlib, shutil
    from peppan_amd import uberBlast as UB, pipeline as PL, clust as CL, synth
    from oracle import oracle as O
    from oracle_context import OracleContext
    names, seqs = synth.make_genes(1000, 1002, seed=355)
    prio = {i: [0, -len(s), i] for i, s in enumerate(seqs)}
    params = dict(noDiamond=nucl, match_identity=0.5, match_frag_len=50, n_thread=2, match_frag_prop=0.25, gtable=11,
                  clust_identity=0.9, clust_match_prop=0.8, incompleteCDS='', match_len=250., match_len1=100., match_len2=400.,
                  match_prop=0.5, match_prop1=0.8, match_prop2=0.4)

    def oracle_fn(fasta, identity, coverage, n_thread):
        recs = CL.readFasta(fasta)
        rep, _ = O.linclust([O.nt_codes(s) for _, s in recs], float(identity), float(coverage))
        return [(recs[r][0], recs[i][0]) for i, r in enumerate(rep.tolist())]

    results = {}
    for tag in ('gpu', 'ora'):
        d = tmp_path / tag
        d.mkdir()
        monkeypatch.chdir(d)
        with open('p.genes', 'w') as f:
            for i, s in enumerate(seqs):
                f.write('>%d\n%s\n' % (i, s.decode()))
        if tag == 'ora':
            octx = OracleContext()
            monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
            monkeypatch.setattr(PL, 'get_context', lambda device=None: octx)
        with contextlib.redirect_stderr(io.StringIO()):
            ex = PL.iterClust('p', 'p.genes', [], dict(identity=0.9, coverage=0.8, n_thread=2, translate=False,
                                                       cluster_fn=oracle_fn if tag == 'ora' else None))
            pairs = PL.get_similar_pairs(ex, prio, dict(params, clust=ex))
        np.save('p.self_bsn.npy', pairs)
        groups = PL.get_gene_group(ex, 'p.self_bsn.npy')
        labels = PL.gene_group_labels(ex, 'p.self_bsn.npy', n_genes=len(seqs))
        results[tag] = (open(ex).read(), np.load('p.clust.npy').tolist(), pairs.tolist(),
                        [[int(k), [int(x) for x in v]] for k, v in groups.items()], labels.tolist())
    g, o = results['gpu'], results['ora']
    assert g[0] == o[0] and g[1] == o[1]          # exemplars, clust.npy
    assert g[2] == o[2]                           # ortholog pairs with their integer identities
    assert g[3] == o[3] and g[4] == o[4]          # gene groups: exact dict and GPU labels
    # the GPU labels describe the same partition as the reference-shaped dict
    part = {}
    for i, l in enumerate(g[4]):
        part.setdefault(l, set()).add(i)
    assert {frozenset(v) for v in part.values() if len(v) > 1} == {frozenset(m) for _, m in g[3]}
    assert len(g[2]) > (150 if nucl else 300) and len(g[3]) > 100


def test_full_size_10k_properties(ctx):
    """BASELINE configs[2] size (10k genes x 1002 nt, all-vs-all): size-independent properties instead of the oracle"""
    from peppan_amd import _native as N, synth, uberBlast as UB
    names, seqs = synth.make_genes(10000, 1002, seed=355)
    ctx.set_query_nt(seqs, 11)
    ctx.set_ref_nt(seqs, 6, 11)
    p = N.default_params(45., 25., 10, 5)
    h1, c1, s1 = ctx.search(p)
    h2, c2, s2 = ctx.search(N.default_params(45., 25., 10, 5, use_lds=0))      # global-memory residue path
    ctx.translate(force=True)
    h3, c3, s3 = ctx.search(p)                                                  # determinism across runs
    for h, c in ((h2, c2), (h3, c3)):
        assert np.array_equal(h1, h) and np.array_equal(c1, c)
    assert len(h1) > 30000 and s1['candidates'] > 40000
    qm, tm = ctx.query_meta(), ctx.target_meta()
    # ordered by (q, t), at most top_k per (q, split)
    key = h1['q'].astype(np.int64) * (1 << 32) + h1['t']
    assert (np.diff(key) > 0).all()
    per = np.bincount(h1['q'].astype(np.int64) * 5 + h1['t'] % 5)
    assert per.max() <= 10
    # CIGAR consistency for every hit
    lens, ops = (c1 >> 2).astype(np.int64), c1 & 3
    owner = np.repeat(np.arange(len(h1)), h1['cigar_runs'])
    assert np.array_equal(np.bincount(owner, weights=lens * (ops != 2), minlength=len(h1)).astype(np.int64), h1['q_end'].astype(np.int64) - h1['q_start'] + 1)
    assert np.array_equal(np.bincount(owner, weights=lens * (ops != 1), minlength=len(h1)).astype(np.int64), h1['t_end'].astype(np.int64) - h1['t_start'] + 1)
    assert np.array_equal(np.bincount(owner, weights=lens, minlength=len(h1)).astype(np.int64), h1['aln_len'].astype(np.int64))
    # every gene finds its own frame-1 translation: one M run over the whole protein, all identities, score = BLOSUM62 self score
    self_hit = (tm['seq'][h1['t']] == h1['q']) & (tm['frame'][h1['t']] == qm['frame'][h1['q']])
    assert self_hit.sum() == 10000
    sh = h1[self_hit]
    # (the final residue is the stop codon 'X', X-X scores -1, so the local alignment ends one residue earlier)
    assert (sh['cigar_runs'] == 1).all() and (sh['n_ident'] == sh['aln_len']).all() and (sh['aln_len'] == qm['aa_len'][sh['q']] - 1).all()
    assert (sh['q_start'] == 1).all() and (sh['t_start'] == 1).all()
    qa, qo = ctx.query_aa()
    diag = np.array(list(p.sub), dtype=np.int64).reshape(32, 32).diagonal()
    cum = np.concatenate([[0], np.cumsum(diag[qa])])
    assert np.array_equal(sh['score'].astype(np.int64), cum[qo[sh['q'] + 1].astype(np.int64) - 1] - cum[qo[sh['q']].astype(np.int64)])
    # union-find: labels are canonical (smallest member), idempotent, and each family of 4 ends up together or split, never mixed
    genes_t = tm['seq'][h1['t']]
    lab = ctx.components(10000, h1['q'], genes_t)
    assert np.array_equal(ctx.components_of_hits(10000, h1, tm['seq']), lab)          # the same graph straight from the hit table
    assert np.array_equal(ctx.components_of_hits(10004, h1, tm['seq'] + 4, q_base=4)[4:], lab + 4)
    # ... and from the table's DEVICE copy (pep_components_of_result): the newest search's, held through search(copy=False)
    h3, c3, s3 = ctx.search(p, copy=False)
    assert np.array_equal(np.array(h3), h1)
    assert np.array_equal(ctx.components_of_search(10000, tm['seq']), lab)
    assert np.array_equal(ctx.components_of_search(10004, tm['seq'] + 4, q_base=4)[4:], lab + 4)
    assert np.array_equal(ctx.components_of_search(10000, tm['seq']), lab)              # (node map changed back: uploaded again)
    # ... and as the tail of the search itself (pep_set_grouping): same labels, with and without a host copy of the table, with a node offset,
    # for a search without hits; switched off again afterwards (a later search carries no labels)
    try:
        ctx.set_grouping(10000, tm['seq'])
        for copy in (False, True):
            h4, c4, s4 = ctx.search(p, copy=copy)
            assert np.array_equal(np.array(h4), h1) and np.array_equal(ctx.labels, lab)
        ctx.set_grouping(10004, tm['seq'] + 4, q_base=4)
        ctx.search(p, copy=False)
        assert np.array_equal(ctx.labels[4:], lab + 4) and np.array_equal(ctx.labels[:4], np.arange(4))
        with pytest.raises(Exception):
            ctx.set_grouping(100, tm['seq'])                                             # a node beyond n_nodes
        ctx.set_grouping(10000, tm['seq'])
        strict = N.default_params(45., 25., 10, 5, max_evalue=1e-300)
        h5, c5, s5 = ctx.search(strict)
        assert len(h5) == 0 and np.array_equal(ctx.labels, np.arange(10000))
    finally:
        ctx.set_grouping(0)
    ctx.search(p, copy=False)
    assert ctx.labels is None
    assert np.array_equal(lab[lab], lab) and (lab <= np.arange(10000)).all()
    assert np.array_equal(ctx.components(10000, np.arange(10000), lab), lab)
    assert (lab // 4 == np.arange(10000) // 4).all()


@pytest.mark.parametrize('gene_len', [1002, 0])
def test_full_size_10k_bit_exact_vs_oracle(ctx, gene_len):
    """gene_len 0: log-normal gene lengths (120 .. 9492 nt, the spread of the reference's example genomes).  gene_len 1002: the headline workload itself (10k genes x 1002 nt all-vs-all, BASELINE configs[2] search stage): every field of every hit and
    the CIGAR arena equal the CPU oracle's.  The oracle gets the proteins K1 produced (K1 has its own golden / oracle tests) and runs
    its OpenMP build on all host cores - a few seconds on the GPU box."""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    names, seqs = synth.make_genes(10000, gene_len, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])
    nts = [seqs[i] for i in order]
    ctx.set_query_nt(nts, 11)
    ctx.set_ref_nt(nts, 6, 11)
    p = N.default_params(45., 25., 10, 5)
    gh, gc, st = ctx.search(p)
    qa, qo = ctx.query_aa()
    ta, to = ctx.target_aa()
    q_aa = [qa[qo[i]:qo[i + 1]] for i in range(len(qo) - 1)]
    t_aa = [ta[to[i]:to[i + 1]] for i in range(len(to) - 1)]
    O.lib().oracle_set_threads(O.granted_cpus())                # all CPUs the box grants (the results do not depend on the thread count)
    oh, oc, ost = O.search(q_aa, t_aa, O.default_params(45., 25., 10, 5))
    _cmp_hits(gh, gc, oh, oc)
    assert len(gh) > 30000
    for k in ('candidates', 'pairs', 'cells', 'tracebacks'):
        assert st[k] == ost[k], k


def _shard_worker(rank, world, port, grid, out_q):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import torch.distributed as dist
    from peppan_amd import _native as N, synth, dist as pdist
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    names, seqs = synth.make_genes(600, 0, seed=91)
    with N.Context(0) as ctx:                     # the ranks share the one GPU of the test box; on a node each has its own
        shard = pdist.ShardedSearch(ctx, seqs, seqs, N.default_params(45., 25., 3, 5), rank, world, grid=grid)
        allh, allc, st = shard.search()
        lab = ctx.components_of_hits(len(seqs), allh, shard.gene_of_target)
    out_q.put((rank, allh.tobytes(), allc.tobytes(), lab.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,grid', [(2, (2, 1)), (2, (1, 2)), (4, (2, 2))])
def test_sharded_search_equals_single_gpu(ctx, world, grid):
    """N>1 path end to end on the GPU: the ranks (gloo transport, one shared GPU here) search the cells of a query x reference grid,
    all-gather their hit tables, merge the top-k, and every rank ends with exactly the single-process table and the same clusters"""
    import socket
    import torch.multiprocessing as mp
    from peppan_amd import _native as N, synth, dist as pdist
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    procs = [mpc.Process(target=_shard_worker, args=(r, world, port, grid, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    names, seqs = synth.make_genes(600, 0, seed=91)
    one = pdist.ShardedSearch(ctx, seqs, seqs, N.default_params(45., 25., 3, 5))
    h, c, st = one.search()
    lab = ctx.components_of_hits(len(seqs), h, one.gene_of_target)
    for rank, hb, cb, lb in res:
        gh = np.frombuffer(hb, dtype=N.HIT_DTYPE)
        assert np.array_equal(gh, h) and np.array_equal(np.frombuffer(cb, dtype=np.uint32), c)
        assert np.array_equal(np.frombuffer(lb, dtype=np.uint32), lab)
    assert len(h) > 1000


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent spawns the ranks (both on device 0 here, gloo exchange), relays ONE JSON line
    with n_gpus 2, and the gathered + merged table equals the CPU oracle's"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PEPPAN_BENCH_SHARE_GPU='1', PEPPAN_BENCH_DETAIL=str(tmp_path / 'detail.json'))
    env.pop('RANK', None); env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--genes', '400', '--steps', '2', '--warmup', '1', '--no-e2e'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['steps'] == 2
    assert line['parity_check']['gpu_hits_identical'] is True and line['parity_check']['hits_compared'] > 400
    assert line['cpu_baseline'] is None                         # the CPU baseline figure belongs to the N = 1 line
    one = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--genes', '400', '--steps', '2', '--warmup', '1'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    assert one.stdout.decode().count('\n') == 1 and len(one.stdout) < 4096          # ONE compact line on stdout, the rest in the detail file
    l1 = json.loads(one.stdout.decode())
    assert l1['roofline']['bound'] == 'hbm' and l1['roofline']['frac'] > 0 and l1['cpu_baseline']['kind'] == 'port' and l1['detail'] == 'detail.json'
    l1 = json.load(open(str(tmp_path / 'detail.json')))
    assert l1['n_gpus'] == 1 and l1['cpu_baseline']['gpu_hits_identical'] is True and l1['hits_per_step'] == line['hits_per_step']
    assert l1['clusters'] == line['clusters'] and l1['uberblast_e2e_ms'] > 0 and l1['ms_per_step_incl_h2d'] > 0
    assert len(l1['roofline_kernels']) == 3 and l1['roofline']['kernel'] == l1['roofline_kernels'][0]['kernel']
    assert l1['cpu_baseline']['kind'] == 'port' and set(l1['cpu_baseline']['reference_binaries']) == {'diamond', 'blastn', 'makeblastdb', 'mmseqs'}
    assert l1['cpu_baseline']['phase_s']['seed'] > 0 and l1['cpu_baseline']['sw_cells_per_s_vectorised'] > l1['cpu_baseline']['sw_cells_per_s']
    # --grid: the reference side split too, so that the exact top-k merge runs behind the exchange; the line says how many ranks the collective connected
    grid = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--grid', '1x2', '--genes', '400', '--steps', '2', '--warmup', '1', '--no-e2e'],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert grid.returncode == 0, grid.stderr.decode()[-2000:]
    lg = json.loads([l for l in grid.stdout.decode().splitlines() if l.startswith('{')][0])
    assert lg['rccl_ranks_seen'] == 2 and 'grid 1 query shards x 2 reference shards' in lg['config']['parallelism']
    assert lg['parity_check']['gpu_hits_identical'] is True and lg['hits_per_step'] == line['hits_per_step'] and lg['clusters'] == line['clusters']
    # the mapping stage as strong scaling: ONE fixed set of genomes dealt to the two ranks, rank 0 writes the stores
    ms = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'map', '--map-scaling', 'strong', '--genes', '600', '--map-genomes', '5'],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert ms.returncode == 0, ms.stderr.decode()[-2000:]
    lm = json.loads([l for l in ms.stdout.decode().splitlines() if l.startswith('{')][0])
    assert lm['n_gpus'] == 2 and lm['scaling'] == 'strong' and lm['value'] > 0 and lm['rccl_ranks_seen'] == 2 and '5 genomes' in lm['config']['workload']
    # ... and with every rank dealing ITS genomes to two worker processes of its own (each with a HIP context on the rank's device)
    mw = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'map', '--map-scaling', 'strong', '--genes', '600', '--map-genomes', '9', '--map-workers', '2'],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert mw.returncode == 0, mw.stderr.decode()[-2000:]
    lw = json.loads([l for l in mw.stdout.decode().splitlines() if l.startswith('{')][0])
    assert lw['n_gpus'] == 2 and lw['workers_per_rank'] == 2 and lw['value'] > 0 and '9 genomes' in lw['config']['workload']


def test_multiple_hsps_per_subject(ctx):
    """hsp_mode 1 (nucleotide tool): a contig carrying two diverged copies of a gene plus an adjacent-band duplicate case"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    names, seqs = synth.make_genes(60, 0, seed=23)
    codes = [O.nt_codes(s) for s in seqs]
    rc = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    sp = lambda n: rng.integers(0, 4, n).astype(np.uint8)
    def mutated(c, rate):
        c = c.copy(); m = rng.random(len(c)) < rate; c[m] = rng.integers(0, 4, int(m.sum())); return c
    contig = np.concatenate([sp(700), codes[0], sp(333), mutated(codes[0], 0.04), sp(90), rc[codes[4][::-1]], sp(2000), mutated(codes[4], 0.08), sp(50)])
    targets = [contig, rc[contig[::-1]]] + codes[:10]
    p = N.nucleotide_params(70., 25.)
    assert p.hsp_mode == 1
    ctx.set_query_aa(codes)
    ctx.set_ref_aa(targets)
    gh, gc, st = ctx.search(p)
    ms = np.array([O.min_score(len(c), p.dbsize, p.max_evalue, p.ka_lambda, p.ka_k) for c in codes], dtype=np.int32)
    oh, oc, ost = O.search(codes, targets, O.params_from(p), min_scores=ms)
    _cmp_hits(gh, gc, oh, oc)
    assert st['tracebacks'] == ost['tracebacks']
    # both copies of gene 0 on the forward strand of the contig, both copies of gene 4 (one per strand)
    assert ((gh['q'] == 0) & (gh['t'] == 0)).sum() == 2
    assert ((gh['q'] == 4) & (gh['t'] == 0)).sum() == 1 and ((gh['q'] == 4) & (gh['t'] == 1)).sum() == 1
    # the same search with one alignment per (q, t) keeps only the better copy
    p0 = N.nucleotide_params(70., 25.); p0.hsp_mode = 0
    g0, c0, s0 = ctx.search(p0)
    assert ((g0['q'] == 0) & (g0['t'] == 0)).sum() == 1 and len(g0) < len(gh)
    o0, oc0, _ = O.search(codes, targets, O.params_from(p0), min_scores=ms)
    _cmp_hits(g0, c0, o0, oc0)


def test_c_abi_error_conventions():
    """errors are loud and carry a message; nothing is truncated silently (include/peppan_hip.h)"""
    from peppan_amd import _native as N
    with N.Context(0) as c:
        with pytest.raises(N.PepError, match='before both sequence sets'):
            c.search(N.default_params())
        c.set_query_aa([np.zeros(10, np.uint8)])
        with pytest.raises(N.PepError):
            c.target_meta()
        c.set_ref_aa([np.zeros(10, np.uint8)])
        bad = N.default_params(); bad.top_k = 0
        with pytest.raises(N.PepError, match='invalid search parameters'):
            c.search(bad)
        bad = N.default_params(); bad.base = 40
        with pytest.raises(N.PepError):
            c.search(bad)
        bad = N.default_params(); bad.xdrop = 99
        with pytest.raises(N.PepError, match='ungapped'):
            c.search(bad)
        bad = N.default_params(); bad.hsp_mode = 7
        with pytest.raises(N.PepError, match='hsp_mode'):
            c.search(bad)
        bad = N.default_params(); bad.gap_ext = 0
        with pytest.raises(N.PepError, match='gap costs'):
            c.search(bad)
        bad = N.default_params(); bad.sub[31 * 32 + 3] = 5
        with pytest.raises(N.PepError, match='padding code'):
            c.search(bad)
        with pytest.raises(N.PepError, match='frames'):
            c.set_ref_nt([b'ACGT'], frames=5)
        with pytest.raises(N.PepError, match='out of range'):
            c.components(4, [0, 9], [1, 2])
        with pytest.raises(N.PepError):
            c.linclust([np.zeros(30, np.uint8)], 0.9, 0.9, base=4, k=40)
        # after errors the context still works
        h, cg, st = c.search(N.default_params())
        assert len(h) == 0
    with pytest.raises(N.PepError):
        N.Context(99)


@pytest.mark.parametrize('hsp_mode', ['1', '2'])
def test_run_batch_equals_per_genome_runs(tmp_path, monkeypatch, hsp_mode):
    """GPU-native batching of the genes->genomes mapping (PEPPAN.py:907-922): ONE search per tool over several genomes,
    ranking inside each genome, gives exactly the per-genome uberBlast results (tables and overlaps).  hsp_mode 2: the nucleotide tool's
    BLAST-like culling and its hit list counted in subjects work per genome of a batch too (competition classes x subjects)"""
    import io, contextlib
    from peppan_amd import uberBlast as UB, synth, configure
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('PEPPAN_BLAST_HSP_MODE', hsp_mode)
    rng = np.random.default_rng(12)
    names, seqs = synth.make_genes(80, 0, seed=19, family=4)
    _write_fasta('genes.fa', [str(i) for i in range(len(seqs))], seqs)
    spacer = lambda: bytes(rng.choice(list(b'ACGT'), int(rng.integers(40, 400))).tolist())
    genomes = []
    for g in range(4):
        contigs, cur = [], spacer()
        for k in rng.permutation(len(seqs))[:45]:
            s = seqs[k]
            if rng.random() < 0.5:
                s = configure.rc(s.decode()).encode()
            cur += s + spacer()
            if rng.random() < 0.12:
                contigs.append(cur); cur = spacer()
        contigs.append(cur + seqs[3] + spacer() + seqs[3] + spacer())          # a duplicated gene on one contig
        path = 'genome%d.fa' % g
        _write_fasta(path, ['%d:c%d' % (g, i) for i in range(len(contigs))], contigs)
        genomes.append(path)
    flags = '-f -m -O --blastn --diamond --min_id 0.4 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11'
    with contextlib.redirect_stderr(io.StringIO()):
        single = [UB.uberBlast(('-r %s -q genes.fa ' % p + flags).split()) for p in genomes]
        batch = UB.uberBlastBatch(genomes, ('-q genes.fa ' + flags).split())
        monkeypatch.setattr(UB.RunBlast, 'MAX_BATCH_NT', 130000)       # forces two sub-batches of two genomes
        split = UB.uberBlastBatch(genomes, ('-q genes.fa ' + flags).split())
    assert len(batch) == 4 and len(split) == 4
    for (t1, o1), (t2, o2), (t3, o3) in zip(single, batch, split):
        assert t1.shape[0] > 40 and t1.tolist() == t2.tolist() and o1.tolist() == o2.tolist()
        assert t1.tolist() == t3.tolist() and o1.tolist() == o3.tolist()


def test_k11_overlaps_vs_oracle_and_golden(ctx):
    from oracle import oracle as O
    from peppan_amd import mapfilters
    rng = np.random.default_rng(3)
    for n, span in ((0, 10), (1, 10), (500, 40000), (20000, 2000000)):
        contig = np.sort(rng.integers(0, 7, n)).astype(np.int32)
        start = rng.integers(1, span, n)
        end = start + rng.integers(30, 3000, n)
        order = np.lexsort((end, start, contig))
        contig, start, end = contig[order], start[order], end[order]
        rid = rng.permutation(n)
        for ovl_l, ovl_p in ((300., 0.6), (30., 0.1)):
            got = ctx.overlaps(contig, start, end, rid, ovl_l, ovl_p)
            assert np.array_equal(got, O.overlaps_sweep(contig.tolist(), start.tolist(), end.tolist(), rid.tolist(), ovl_l, ovl_p))
    g = load_golden('g07_filters.json')
    for case in g['cases']:
        tab = np.empty([len(case['table']), 16], dtype=object)
        for i, r in enumerate(case['table']):
            for j, v in enumerate(r):
                tab[i, j] = v
        assert mapfilters.overlaps(tab, 30, 0.1, sweep=ctx.overlaps).tolist() == case['overlap_30_01_raw']
    # unsorted input is an error, not garbage
    from peppan_amd import _native as N
    with pytest.raises(N.PepError, match='sorted'):
        ctx.overlaps([1, 0], [5, 5], [9, 9], [0, 1], 300., 0.6)


def test_get_map_bsn_batched_equals_per_genome_workers(tmp_path, monkeypatch):
    """PEPPAN.py:907-989 on the GPU: get_map_bsn's batched search fills the same four stores as the reference's scheme of one
    iter_map_bsn call per genome (PEPPAN.py:759-867, 922), and every full-length planted gene comes back as a group"""
    import io, contextlib
    from peppan_amd import mapbsn, synth
    monkeypatch.chdir(tmp_path)
    names, seqs = synth.make_genes(120, 0, seed=23, family=4)
    _write_fasta('m.clust.exemplar', [str(i) for i in range(len(seqs))], seqs)
    worlds = synth.make_genomes(seqs, 5, seed=77)
    genomes, old = {}, {}
    for g, (gname, contig, ann) in enumerate(worlds):
        cid = 5000 + g
        genomes[cid] = [900 + g, contig.decode()]
        old[str(cid)] = np.array([[k, s, e, strand, 1] for k, s, e, strand in ann[::2]], dtype=object)    # half of them were annotated
    with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
        for c, v in old.items():
            op.save(c, v)
    np.save('m.self_bsn.npy', np.array([[0, 1, 9000], [4, 5, -2], [8, 9, 0]], dtype=int))
    params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
                  match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)

    def per_genome(prefix, clust, jobs, p):                     # the reference's scheme: one worker call per genome
        for id, taxon, seq in jobs:
            z = np.load(mapbsn.iter_map_bsn((prefix, clust, id, taxon, seq, 'm.self_bsn.npy', 'm.old_prediction.npz', p)) + '.bsn.npz',
                        allow_pickle=True)
            yield z['bsn'], z['ovl']
    stores = {}
    with contextlib.redirect_stderr(io.StringIO()):
        per = list(per_genome('m', 'm.clust.exemplar', [(i, t, [[c, s]]) for i, (c, (t, s)) in enumerate(genomes.items())], params))
        for tag, kw in (('batch', {}), ('batch2', {'search': lambda *a: mapbsn._gpu_search(*a, genomes_per_batch=2)})):
            fn = ['%s.%s.npz' % (tag, x) for x in ('tab', 'seq', 'mat', 'conflicts')]
            with mapbsn.MapBsn(fn[0], 'w') as c0, mapbsn.MapBsn(fn[1], 'w') as c1, mapbsn.MapBsn(fn[2], 'w') as c2, mapbsn.MapBsn(fn[3], 'w') as c3:
                mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', c0, c1, c2, c3, True, params, **kw)
            out = {}
            for x, f in zip(('tab', 'seq', 'mat', 'conflicts'), fn):
                with mapbsn.MapBsn(f) as c:
                    out[x] = {k: c.get(k) for k in sorted(c.keys())}
            stores[tag] = out

    def plain(x):
        if isinstance(x, np.ndarray):
            return [plain(v) for v in x.tolist()]
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        return x
    assert plain(stores['batch']) == plain(stores['batch2'])
    # the stores against the per-genome worker outputs
    mats = [m for k in sorted(stores['batch']['mat'], key=int) for m in stores['batch']['mat'][k]]
    alle = [m for k in sorted(stores['batch']['seq'], key=int) for m in stores['batch']['seq'][k]]
    want_m = [rows for bsn, ovl in per for rows in bsn.T[6]]
    want_s = [a for bsn, ovl in per for a in bsn.T[4]]
    assert len(mats) == len(want_m) > 300 and plain(mats) == plain(want_m) and plain(alle) == plain(want_s)
    tab = np.vstack(list(stores['batch']['tab'].values()))
    assert tab.shape[0] == len(want_m) and sorted(tab.T[5].tolist()) == list(range(len(want_m)))
    # every planted allele (<= 2 % substitutions, full length) is found in its genome with identity >= 0.95
    for g, (gname, contig, ann) in enumerate(worlds):
        found = {(int(r[0]), int(r[1])) for r in tab if r[3] >= 9500}
        for k, s, e, strand in ann:
            assert (k, 900 + g) in found
    # column 10 of the hit rows: a locus the old annotation holds is recognised (>= 0.9 of it covered, in frame)
    best = {}
    for rows in want_m:
        for r in rows:
            key = (r[1], r[0])
            best.setdefault(key, []).append((min(r[8], r[9]), max(r[8], r[9]), r[10]))
    for g, (gname, contig, ann) in enumerate(worlds):
        for i, (k, s, e, strand) in enumerate(ann[::2]):
            hits = [v for lo, hi, v in best.get((5000 + g, k), []) if min(hi, e) - max(lo, s) + 1 >= 0.9 * (e - s + 1)]
            assert hits and max(hits) >= 0.9


def test_map_workers_write_the_same_stores(tmp_path, monkeypatch):
    """the reference's pool of workers (PEPPAN.py:907-923) on one GPU: worker processes with HIP contexts of their own map rounds of genomes and
    make the stores' members; the four stores are those of the process mapping everything itself - whatever the number of workers and the size
    of a round (members of 1 000 groups are made inside rounds, completed across rounds, and left open at the end)"""
    import io, contextlib
    from peppan_amd import mapbsn, synth
    from peppan_amd.mapworkers import MapWorkers
    monkeypatch.chdir(tmp_path)
    names, seqs = synth.make_genes(1500, 0, seed=29, family=3)
    _write_fasta('m.clust.exemplar', [str(i) for i in range(len(seqs))], seqs)
    worlds = synth.make_genomes(seqs, 11, seed=78)
    genomes = {}
    with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
        for g, (gname, contig, ann) in enumerate(worlds):
            genomes[5000 + g] = [900 + g, contig.decode()]
            op.save(5000 + g, np.array([[k, s, e, strand, 1] for k, s, e, strand in ann[::2]], dtype=object))
    np.save('m.self_bsn.npy', np.array([[0, 1, 9000], [4, 5, -2], [8, 9, 0]], dtype=int))
    params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
                  match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)

    def run(tag, save_seq, **kw):
        fn = ['%s.%s.npz' % (tag, x) for x in ('tab', 'seq', 'mat', 'conflicts')]
        with contextlib.redirect_stderr(io.StringIO()):
            with mapbsn.MapBsn(fn[0], 'w') as c0, mapbsn.MapBsn(fn[1], 'w') as c1, mapbsn.MapBsn(fn[2], 'w') as c2, mapbsn.MapBsn(fn[3], 'w') as c3:
                mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', c0, c1, c2, c3, save_seq, params, **kw)
        out = {}
        for x, f in zip(('tab', 'seq', 'mat', 'conflicts'), fn):
            with mapbsn.MapBsn(f) as c:
                out[x] = {k: c.get(k) for k in sorted(c.keys())}
        return out

    def same(a, b):
        assert {x: sorted(v) for x, v in a.items()} == {x: sorted(v) for x, v in b.items()}
        for x in a:
            for k in a[x]:
                u, v = a[x][k], b[x][k]
                if u.dtype != object:
                    assert u.dtype == v.dtype and np.array_equal(u, v), (x, k)
                else:
                    assert len(u) == len(v)
                    for p, q in zip(u, v):
                        assert p.dtype == q.dtype and p.shape == q.shape and p.tolist() == q.tolist(), (x, k)
    want = run('self', True)
    assert sum(len(v) for v in want['mat'].values()) > 3000 and len(want['mat']) >= 4            # several members of 1 000 groups
    with MapWorkers(3) as pool:
        same(want, run('w3', True, workers=pool, genomes_per_round=2))
        same(want, run('w3b', True, workers=pool, genomes_per_round=64))          # one round
        no_seq = run('w3c', False, workers=pool, genomes_per_round=3)
        assert no_seq['seq'] == {}
        same({k: v for k, v in want.items() if k != 'seq'}, {k: v for k, v in no_seq.items() if k != 'seq'})
    same(want, run('w2', True, workers=2, genomes_per_round=1))



def _random_loci(rng, n_groups, contig_len=20000, n_contigs=3):
    """random K12 input: contigs with N runs and planted stops, groups of 1-3 rows with indels, both strands"""
    from peppan_amd import _native as N
    contigs = []
    for c in range(n_contigs):
        s = rng.choice(list(b'ACGT'), contig_len).astype(np.uint8)
        for _ in range(6):
            p = int(rng.integers(0, contig_len - 40)); s[p:p + int(rng.integers(1, 30))] = ord('N')
        for p in rng.integers(0, contig_len - 3, 300):
            s[p:p + 3] = list(rng.choice([b'TAA', b'TAG', b'TGA']))
        contigs.append(s.tobytes())
    rows, cigar, grp_off, grp_qlen = [], [], [0], []
    for g in range(n_groups):
        ql = int(rng.integers(60, 2500))
        q_at = 1
        for k in range(int(rng.choice([1, 1, 1, 2, 3]))):
            if q_at > ql - 30:
                break
            qs = max(1, q_at - int(rng.integers(0, 20)) * (k > 0))
            budget = ql - qs + 1
            runs, q_used, r_used = [], 0, 0
            while q_used < budget:
                m = int(min(budget - q_used, rng.integers(1, 400)))
                runs.append((m, 0)); q_used += m; r_used += m
                if q_used >= budget or rng.random() < 0.3:
                    break
                op = int(rng.choice([1, 2])); n = int(rng.integers(1, 8))
                if op == 1:
                    n = min(n, budget - q_used)
                    if n == 0:
                        break
                    q_used += n
                else:
                    r_used += n
                runs.append((n, op))
            if runs[-1][1] == 2:
                r_used -= runs[-1][0]; runs.pop()
            c = int(rng.integers(0, n_contigs))
            lo = int(rng.integers(1, contig_len - r_used))
            rs, re_ = (lo, lo + r_used - 1) if rng.random() < 0.5 else (lo + r_used - 1, lo)
            rows.append((c, qs, rs, re_, len(runs), g, len(cigar)))
            cigar += [(n << 2) | op for n, op in runs]
            q_at = qs + q_used
        grp_off.append(len(rows)); grp_qlen.append(ql)
    return contigs, np.array(rows, dtype=N.LOCUS_DTYPE), np.array(cigar, dtype=np.uint32), np.array(grp_off, dtype=np.uint64), np.array(grp_qlen, dtype=np.uint32)


def test_k12_alleles_vs_oracle(ctx):
    from oracle import oracle as O
    from peppan_amd import _native as N
    rng = np.random.default_rng(1212)
    for n_groups, gtable in ((1, 11), (7, 4), (600, 11), (5000, 4)):
        contigs, rows, cigar, grp_off, grp_qlen = _random_loci(rng, n_groups)
        want = O.alleles(contigs, rows, cigar, grp_off, grp_qlen, gtable)
        got = ctx.alleles(contigs, rows, cigar, grp_off, grp_qlen, gtable)
        for w, g in zip(want, got):
            assert w.shape == g.shape and np.array_equal(w, g)
        assert want[1].min() >= 0 and (want[0] > 0).all()
    # nothing to do / inconsistent input is an error, not a silent truncation
    e = ctx.alleles([b'ACGT'], np.zeros(0, N.LOCUS_DTYPE), np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint32))
    assert all(len(x) == 0 for x in e)
    contigs, rows, cigar, grp_off, grp_qlen = _random_loci(rng, 5)
    bad = rows.copy(); bad['re'][0] += 1
    with pytest.raises(N.PepError, match='inconsistent'):
        ctx.alleles(contigs, bad, cigar, grp_off, grp_qlen)
    bad = rows.copy(); bad['contig'][0] = 99
    with pytest.raises(N.PepError, match='out of range'):
        ctx.alleles(contigs, bad, cigar, grp_off, grp_qlen)


def test_k12_build_bsn_matches_reference_golden(ctx, tmp_path):
    """the GPU allele kernel inside build_bsn reproduces the bsn arrays the reference's own iter_map_bsn wrote (golden G14)"""
    import copy
    from conftest import load_golden
    from peppan_amd import mapbsn
    g = load_golden('g14_mapbsn.json')
    old_fn = str(tmp_path / 'old.npz')
    with mapbsn.MapBsn(old_fn, 'w') as op:
        for contig, rows in g['old_prediction'].items():
            op.save(contig, np.array(rows, dtype=object))

    def plain(x):
        if isinstance(x, np.ndarray):
            return [plain(v) for v in x.tolist()]
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        return x
    for case in g['cases']:
        tab = np.empty([len(case['table']), 17], dtype=object)
        for i, r in enumerate(case['table']):
            for j, v in enumerate(r):
                tab[i, j] = copy.deepcopy(v)
        seq = [[int(c), s] for c, s in case['contigs'].items()]
        bsn, ovl = mapbsn.build_bsn(tab, np.array(case['overlap'], dtype=int).reshape(-1, 3), seq, np.array(g['self_bsn'], dtype=int), old_fn,
                                    dict(g['params']), ctx=ctx)
        assert plain(ovl) == case['ovl'] and plain(bsn) == case['bsn']


def test_k13_sha1_and_dedup(ctx, tmp_path):
    """K13: SHA-1 per gene == hashlib (every padding case), duplicate collapse == the loop of writeGenes, writeGenes over the GPU == golden G12"""
    import hashlib
    from conftest import load_golden
    from oracle import oracle as O
    from peppan_amd import pipeline as PL
    rng = np.random.default_rng(13)
    lens = list(range(0, 140)) + [255, 256, 257, 1000, 1002, 9492, 100000] + [int(x) for x in rng.integers(1, 4000, 300)]
    seqs = [bytes(rng.choice(list(b'ACGTN'), n).tolist()) for n in lens]
    got = ctx.sha1(seqs)
    assert got.shape == (len(seqs), 20)
    for s, d in zip(seqs, got):
        assert d.tobytes() == hashlib.sha1(s).digest()
    assert PL.gene_hashes([s.decode() for s in seqs[:50]], ctx=ctx) == [int(hashlib.sha1(s).hexdigest(), 16) for s in seqs[:50]]
    assert ctx.sha1([]).shape == (0, 20) and len(ctx.dedup([], np.zeros((0, 20), np.uint8))) == 0
    # dedup: few distinct sequences, lengths arranged in runs that re-open
    for n, n_distinct in ((1, 1), (50, 4), (20000, 300), (300000, 5000)):
        pool = [bytes(rng.choice(list(b'ACGT'), int(rng.choice([30, 30, 33, 36]))).tolist()) for _ in range(n_distinct)]
        pick = rng.integers(0, n_distinct, n)
        if n > 100:                                      # mostly length-sorted like PEPPAN's priority, with a few runs out of place
            pick = pick[np.argsort([-len(pool[k]) for k in pick], kind='stable')]
            cut = n // 3
            pick = np.concatenate([pick[cut:], pick[:cut]])
        lengths = np.array([len(pool[k]) for k in pick], dtype=np.uint32)
        dig = O.sha1_digests(pool)[pick]
        want = O.dedup(lengths, dig)
        assert np.array_equal(ctx.dedup(lengths, dig), want)
        assert (want <= np.arange(n)).all()
    g = load_golden('g12_writegenes.json')
    genes = {int(k): ['f', '', 0, 0, '+', v[0], v[1]] for k, v in g['genes'].items()}
    prio = {int(k): v for k, v in g['priority'].items()}
    fn, groups = PL.writeGenes(str(tmp_path / 'w.genes'), genes, prio, ctx=ctx)
    assert open(fn).read() == g['fasta'] and groups == g['groups']


def test_results_stay_valid_across_searches(ctx):
    """a pep_result handle is independent of later calls on its context: its table lives in the context's pinned staging area only
    until the next search, which first gives the older result its own copy (include/peppan_hip.h: ownership)"""
    import ctypes as C
    from peppan_amd import _native as N, synth
    lib = N.load_library()
    names, seqs = synth.make_genes(300, 0, seed=77)
    ctx.set_query_nt(seqs[:150], 11); ctx.set_ref_nt(seqs, 6, 11)
    p = N.default_params(45., 25., 10, 5)
    want_h, want_c, _ = ctx.search(p)
    r1, r2 = C.c_void_p(), C.c_void_p()
    assert lib.pep_search(ctx._h, C.byref(p), C.byref(r1)) == 0
    ctx.set_query_nt(seqs[150:], 11)                                   # different second search; r1 not copied out yet
    assert lib.pep_search(ctx._h, C.byref(p), C.byref(r2)) == 0
    ctx.translate(force=True)                                          # K1 staging is a different buffer, but exercise it anyway
    for r in (r1, r2):
        nh, nc = C.c_uint64(), C.c_uint64()
        assert lib.pep_result_size(r, C.byref(nh), C.byref(nc)) == 0
        h, c = np.zeros(nh.value, dtype=N.HIT_DTYPE), np.zeros(max(nc.value, 1), dtype=np.uint32)
        assert lib.pep_result_copy(r, h.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)) == 0
        if r is r1:
            assert nh.value == len(want_h) > 50 and h.tobytes() == want_h.tobytes() and c[:nc.value].tobytes() == want_c.tobytes()
        else:
            assert nh.value > 50 and h.tobytes() != want_h.tobytes()
    lib.pep_result_free(r2)
    lib.pep_result_free(r1)
    again, _, _ = ctx.search(p)                                        # freeing a staged result leaves the context usable
    assert len(again) == nh.value


def test_k9_gapped_verification_and_context_state(ctx):
    """alleles that differ from their centre by an indel are absorbed through the alignment engine (== oracle), and a context
    keeps working for searches afterwards: nucleotide inputs are re-translated, amino-acid inputs must be given again"""
    from oracle import oracle as O
    from peppan_amd import _native as N, synth
    rng = np.random.default_rng(99)
    fam = []
    for f in range(60):
        centre = rng.integers(0, 4, int(rng.integers(300, 2500))).astype(np.uint8)
        fam.append(centre)
        for v in range(int(rng.integers(1, 5))):
            x = centre.copy()
            for p in rng.choice(len(x), max(1, len(x) // 60), replace=False):
                x[p] = (x[p] + 1) % 4
            cut = int(rng.integers(20, len(x) - 40))
            if rng.random() < 0.5:
                x = np.concatenate([x[:cut], x[cut + int(rng.integers(1, 13)):]])
            else:
                x = np.concatenate([x[:cut], rng.integers(0, 4, int(rng.integers(1, 13))).astype(np.uint8), x[cut:]])
            fam.append(x)
    long_pair = rng.integers(0, 4, 9400).astype(np.uint8)                      # beyond the packed 16-bit range of the DP
    fam += [long_pair, np.concatenate([long_pair[:4000], long_pair[4003:]])]
    names, seqs = synth.make_genes(200, 0, seed=5)
    ctx.set_query_nt(seqs[:50], 11); ctx.set_ref_nt(seqs, 6, 11)
    p = N.default_params(45., 25., 10, 5)
    before, _, _ = ctx.search(p)
    for min_id, cov in ((0.95, 0.9), (0.9, 0.5)):
        g_rep, g_st = ctx.linclust(fam, min_id, cov)
        o_rep, o_st = O.linclust(fam, min_id, cov)
        assert np.array_equal(g_rep, o_rep) and g_st == o_st
        assert len(set(g_rep.tolist())) <= 70 and g_rep[-1] == g_rep[-2]
    after, _, _ = ctx.search(p)                                               # nucleotide inputs: K1 runs again by itself
    assert before.tobytes() == after.tobytes() and len(before) > 20
    ctx.set_query_aa([np.arange(20, dtype=np.uint8)] * 3); ctx.set_ref_aa([np.arange(20, dtype=np.uint8)] * 3)
    ctx.linclust(fam[:10], 0.9, 0.9)
    with pytest.raises(N.PepError):
        ctx.search(p)


@pytest.mark.parametrize('seed', [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_search_parameter_fuzz(ctx, seed):
    """random gap costs, ungapped-filter settings, top-k / splits, HSP mode and thresholds: GPU == oracle on every field;
    then the packed 16-bit passes against the 32-bit passes of the GPU itself on a larger set"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    rng = np.random.default_rng(1000 + seed)
    prots = synth.make_proteins(180, length=(40, 520), seed=50 + seed, family=int(rng.integers(2, 6)), sub=float(rng.uniform(0.05, 0.4)))
    p = N.default_params(float(rng.choice([0., 30., 60.])), float(rng.choice([0., 20., 70.])), int(rng.choice([1, 3, 10])), int(rng.choice([1, 5])),
                         ungapped_min=int(rng.choice([0, 30, 45, 60])))
    p.gap_open, p.gap_ext = int(rng.integers(5, 15)), int(rng.integers(1, 4))
    p.xdrop, p.ext_right, p.ext_left = int(rng.integers(8, 21)), int(rng.integers(8, 49)), int(rng.integers(0, 41))      # (ext_right below 16: the first filter stage judges a shorter extension)
    p.stage1_min = min(int(rng.choice([0, 10, 24, 40])), p.ungapped_min) if p.ungapped_min else int(rng.choice([0, 24]))
    p.hsp_mode = int(rng.integers(0, 2))
    p.max_evalue = float(rng.choice([1., 1e-3, 10.]))
    if seed % 3 == 0:                            # a steeper substitution table (x3): the 16-bit passes must size their eligibility from it
        for x in range(1024):
            if x // 32 != 31 and x % 32 != 31:   # (row / column 31 = the padding code: stays at -64, the library checks it)
                p.sub[x] = max(-128, int(p.sub[x]) * 3)
    nq = int(rng.integers(1, len(prots)))
    ctx.set_query_aa(prots[:nq]); ctx.set_ref_aa(prots)
    gh, gc, st = ctx.search(p)
    ms = np.array([N.min_score(len(s), p.dbsize, p.max_evalue) for s in prots[:nq]], dtype=np.int32)
    oh, oc, ost = O.search(prots[:nq], prots, O.params_from(p), min_scores=ms)
    assert st['candidates'] == ost['candidates'] and st['pairs'] == ost['pairs'] and st['cells'] == ost['cells'] and st['tracebacks'] == ost['tracebacks']
    _cmp_hits(gh, gc, oh, oc)
    # packed (two candidates per wavefront, 16-bit) vs 32-bit passes
    names, seqs = synth.make_genes(1500, 0, seed=200 + seed)
    ctx.set_query_nt(seqs[:700], 11); ctx.set_ref_nt(seqs, 6, 11)
    q = N.default_params(40., 20., 10, 5)
    q.gap_open, q.gap_ext, q.hsp_mode = p.gap_open, p.gap_ext, p.hsp_mode
    a_h, a_c, a_st = ctx.search(q)
    q.reserved[1] = 1
    b_h, b_c, b_st = ctx.search(q)
    assert len(a_h) > 500 and a_h.tobytes() == b_h.tobytes() and a_c.tobytes() == b_c.tobytes()
    assert a_st['cells'] == b_st['cells'] and a_st['tracebacks'] == b_st['tracebacks']


def _real_fixture(tmp_path):
    """golden G16: 1 644 real genes of the reference's examples/ and a 300 kb piece of one chromosome"""
    import gzip
    from conftest import load_golden
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    g = load_golden('g16_real.json')
    seqs = {}
    with gzip.open(os.path.join(here, 'g16_real_genes.fa.gz'), 'rt') as f:
        for line in f:
            if line.startswith('>'):
                cur = int(line[1:])
            else:
                seqs[cur] = line.strip()
    contig = ''.join(l.strip() for l in gzip.open(os.path.join(here, 'g16_real_contig.fa.gz'), 'rt') if not l.startswith('>'))
    return g, seqs, contig


def test_real_genes_on_the_gpu(ctx, tmp_path, monkeypatch):
    """real sequences instead of synthetic ones (variable lengths, paralogs, low-complexity stretches): front end, clustering,
    all-vs-all search and genome mapping on the GPU against the reference's golden values / the oracle-driven host code"""
    import io, contextlib, hashlib
    from peppan_amd import uberBlast as UB, pipeline as PL, clust as CL
    from oracle import oracle as O
    from oracle_context import OracleContext
    monkeypatch.chdir(tmp_path)
    g, seqs, contig = _real_fixture(tmp_path)
    ids = sorted(seqs)
    # K13 on real genes: sha1 codes and duplicate groups are the reference's
    assert PL.gene_hashes([seqs[i] for i in ids], ctx=ctx) == [int(g['hash'][str(i)]) for i in ids]
    genes = {i: ['f', '', 0, 0, '+', int(g['hash'][str(i)]), seqs[i]] for i in ids}
    prio = {int(k): [v[0], v[1], int(v[2])] for k, v in g['priority'].items()}
    fn, groups = PL.writeGenes('real.genes', genes, prio, ctx=ctx)
    assert groups == g['groups'] and [int(l[1:]) for l in open(fn) if l.startswith('>')] == g['unique_order']
    # clustering of the unique genes: GPU clusterer == oracle relation through the same round logic
    def oracle_fn(fasta, identity, coverage, n_thread):
        recs = CL.readFasta(fasta)
        rep, _ = O.linclust([O.nt_codes(s) for _, s in recs], float(identity), float(coverage))
        return [(recs[r][0], recs[i][0]) for i, r in enumerate(rep.tolist())]
    out = {}
    for tag, cfn in (('gpu', None), ('ora', oracle_fn)):
        with contextlib.redirect_stderr(io.StringIO()):
            ex = PL.iterClust(tag, 'real.genes', [[0, 999999, 10000]], dict(identity=0.9, coverage=0.8, n_thread=1, translate=False, cluster_fn=cfn))
        out[tag] = (open(ex).read(), open(tag + '.clust.tab').read(), np.load(tag + '.clust.npy').tolist())
    assert out['gpu'] == out['ora']
    n_ex = out['gpu'][0].count('>')
    assert 300 < n_ex < 1301                          # the four genomes share most genes: alleles collapse onto exemplars
    # all-vs-all of the exemplars and mapping onto the chromosome piece: whole tables, GPU vs the same host code over the oracle
    with open('genome.fa', 'w') as f:
        f.write('>900001\n%s\n' % contig)
    self_argv = '-r gpu.clust.exemplar -q gpu.clust.exemplar --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 2 --min_ratio 0.25 -e 3,3 -p --gtable 11'.split()
    map_argv = '-r genome.fa -q gpu.clust.exemplar -f -m -O --blastn --diamond --min_id 0.55 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11'.split()
    with contextlib.redirect_stderr(io.StringIO()):
        gpu_self = UB.uberBlast(self_argv)
        gpu_map, gpu_ovl = UB.uberBlast(map_argv)
        octx = OracleContext()
        monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
        ora_self = UB.uberBlast(self_argv)
        ora_map, ora_ovl = UB.uberBlast(map_argv)
    assert gpu_self.shape[0] > n_ex and gpu_self.tolist() == ora_self.tolist()
    assert gpu_map.shape[0] > 150 and gpu_map.tolist() == ora_map.tolist() and gpu_ovl.tolist() == ora_ovl.tolist()


def test_many_searches_on_one_context(ctx):
    """a long-lived context: the single-launch scans tag their status words with a per-call epoch (14 bits for the 64-bit scans) and
    clear the state when it wraps - 6 500 searches (three such scans each) pass that point; every result must equal the first"""
    from peppan_amd import _native as N, synth
    prots = synth.make_proteins(48, length=(60, 200), seed=21, family=4, sub=0.2)
    ctx.set_query_aa(prots); ctx.set_ref_aa(prots)
    p = N.default_params(0., 0., 10, 5)
    h0, c0, st0 = ctx.search(p)
    assert len(h0) > 40
    for it in range(6500):
        h, c, st = ctx.search(p, copy=False)
        if it % 500 == 0 or it > 5300:
            assert h.tobytes() == h0.tobytes() and c.tobytes() == c0.tobytes(), it
    h, c, st = ctx.search(p)
    assert h.tobytes() == h0.tobytes() and c.tobytes() == c0.tobytes()


def test_searches_in_flight_together_on_contexts_of_their_own(ctx):
    """two host threads, a context (stream, work space, counters) each, searching different sets at the same time: every result equals the one the
    context produces alone - nothing of a search lives outside its context (what bench.py's `two_searches_in_flight` and a caller with
    several independent searches rely on)"""
    import threading
    from peppan_amd import _native as N, synth
    sets = [synth.make_genes(n, 0, seed=sd)[1] for n, sd in ((700, 5), (1100, 6))]
    p = N.default_params(45., 25., 10, 5)
    ctxs, alone = [ctx, N.Context(0)], []
    try:
        for c, nts in zip(ctxs, sets):
            c.set_query_nt(nts, 11); c.set_ref_nt(nts, 6, 11)
            h, cg, st = c.search(p)
            alone.append((h.tobytes(), cg.tobytes()))
            assert len(h) > 1000
        assert alone[0] != alone[1]
        bad, go = [], threading.Barrier(2)

        def run(k):
            go.wait()
            for it in range(40):
                ctxs[k].invalidate_translation()              # K1 inside the search, like the bench step
                h, cg, st = ctxs[k].search(p, copy=False)
                if (h.tobytes(), cg.tobytes()) != alone[k]:
                    bad.append((k, it))
        th = [threading.Thread(target=run, args=(k,)) for k in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not bad, bad
    finally:
        ctxs[1].close()


def test_query_index_partition_build_equals_plain_build(ctx):
    """the query seed index built by partition (coarse buckets in LDS, per-bucket LDS sort) gives the same search as the count -> scan
    -> fill build (params.reserved[2] = 1), also when a coarse bucket overflows LDS and the library falls back by itself"""
    from peppan_amd import _native as N, synth
    names, seqs = synth.make_genes(2500, 0, seed=91)
    rep = b'ATG' + b'GCTGCTGCAGCA' * 250 + b'TAA'                  # 1 000 alanines: one k-mer, a thousand times per gene, in 60 genes
    cases = [(seqs, seqs), (list(seqs[:300]) + [rep] * 60, list(seqs[:600]) + [rep] * 2)]
    for qs, ts in cases:
        ctx.set_query_nt(qs, 11); ctx.set_ref_nt(ts, 6, 11)
        out = []
        for flag in (0, 1):
            p = N.default_params(45., 25., 10, 5)
            p.reserved[2] = flag
            h, c, st = ctx.search(p)
            out.append((h.tobytes(), c.tobytes(), st['query_seeds'], st['target_seeds'], st['seed_hits'], st['candidates'], st['pairs']))
        assert out[0] == out[1] and len(out[0][0]) > 64 * 100


def test_sensitive_mode_four_shapes_vs_oracle(ctx):
    """default_params(sensitive=True): four seed shapes through K2-K4 (one index build + join per shape into one candidate set), every
    field equal to the oracle run with the same parameter block"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    names, seqs = synth.make_genes(1200, 0, seed=19)
    ctx.set_query_nt(seqs, 11)
    ctx.set_ref_nt(seqs, 6, 11)
    p = N.default_params(45., 25., 10, 5, sensitive=True)
    assert p.n_shapes == 4
    gh, gc, st = ctx.search(p)
    base_h, _, base_st = ctx.search(N.default_params(45., 25., 10, 5))
    qa, qo = ctx.query_aa()
    ta, to = ctx.target_aa()
    q_aa = [qa[qo[i]:qo[i + 1]] for i in range(len(qo) - 1)]
    t_aa = [ta[to[i]:to[i + 1]] for i in range(len(to) - 1)]
    oh, oc, ost = O.search(q_aa, t_aa, O.params_from(p))
    _cmp_hits(gh, gc, oh, oc)
    assert st['candidates'] == ost['candidates'] and st['candidates'] >= base_st['candidates'] and len(gh) >= len(base_h) > 3000


def test_alignment_stage_without_host_round_trips_equals_exactly_sized_stage(ctx):
    """the alignment stage runs from the candidate count to the result sizes without a host round trip (the number of selected pairs stays
    on the device, buffers sized from upper bounds); params.reserved2 = 1 makes it synchronise after the selection and size everything
    exactly.  Both give the same table and the same statistics - protein families, the nucleotide configuration (hsp_mode 1, long
    pairs), the 32-bit passes, a search whose candidates all die at the score cut, and K1 with its host tables built on demand"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    keys = ('candidates', 'pairs', 'tracebacks', 'tracebacks_gapless', 'hits', 'cells', 'cells_swept', 'cells_trace', 'cells_swept_trace', 'dir_bytes')
    prots = synth.make_proteins(300, length=(40, 500), seed=23, family=5, sub=0.25)
    names, seqs = synth.make_genes(120, 0, seed=4)
    codes = [O.nt_codes(s.decode()) for s in seqs]
    rc = [(3 - c[::-1]).astype(np.uint8) for c in codes]
    rng = np.random.default_rng(5)
    aa = np.frombuffer(b'ARNDCQEGHILKMFPSTWYV', dtype=np.uint8) - 65
    noise = [aa[rng.integers(0, 20, 150)] for _ in range(40)]
    cases = [(prots, prots, N.default_params(30., 10., 10, 5), 300), (codes, codes + rc, N.nucleotide_params(60., 20.), 120),
             (noise, noise[::-1], N.default_params(0., 0., 10, 5, max_evalue=1e-30), 0)]
    p32 = N.default_params(30., 10., 3, 5)
    p32.reserved[1] = 1
    cases.append((prots, prots, p32, 300))
    for qs, ts, p, least in cases:
        ctx.set_query_aa(qs); ctx.set_ref_aa(ts)
        out = []
        for flag in (0, 1):
            p.reserved2 = flag
            h, c, st = ctx.search(p)
            out.append((h.tobytes(), c.tobytes(), tuple(st[k] for k in keys)))
        assert out[0] == out[1], (out[0][2], out[1][2])
        assert len(out[0][0]) >= 64 * least
    names, seqs = synth.make_genes(900, 0, seed=77)
    ctx.set_query_nt(seqs[:400], 11); ctx.set_ref_nt(seqs, 6, 11)
    out = []
    for flag in (0, 1):
        p = N.default_params(45., 25., 10, 5)
        p.reserved2 = flag
        h, c, st = ctx.search(p)
        tm = ctx.target_meta()
        out.append((h.tobytes(), c.tobytes(), tuple(st[k] for k in keys), tm.tobytes()))
    assert out[0] == out[1] and len(out[0][0]) > 64 * 400
    qa, qo = ctx.query_aa()
    ta, to = ctx.target_aa()
    oh, oc, ost = O.search([qa[qo[i]:qo[i + 1]] for i in range(len(qo) - 1)], [ta[to[i]:to[i + 1]] for i in range(len(to) - 1)], O.default_params(45., 25., 10, 5))
    _cmp_hits(np.frombuffer(out[0][0], dtype=N.HIT_DTYPE), np.frombuffer(out[0][1], dtype=np.uint32), oh, oc)


def test_candidate_sort_two_level_and_fallback(ctx):
    """the candidate list is sorted in two levels (top 10 bits of the key, then every bucket inside LDS) unless a bucket holds more than
    4 096 keys - then by four LSD passes; which one runs is decided from a histogram that set_compact computes.  A reference in which one
    query family has 6 000 members overflows its bucket; a plain family set does not; both tables equal the oracle's, and params.reserved[2]
    (LSD path forced) changes nothing"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    rng = np.random.default_rng(12)
    base = synth.make_proteins(24, length=(150, 260), seed=31, family=1, sub=0.)
    crowd = []
    for k in range(6000):                                  # 6 000 mutants of protein 0: one query, thousands of candidates
        p = base[0].copy()
        idx = rng.integers(0, len(p), 6)
        p[idx] = rng.integers(0, 20, 6).astype(np.uint8)
        crowd.append(p)
    for qs, ts, top_k in ((base, crowd + base, 50), (base, base, 10)):
        ctx.set_query_aa(qs); ctx.set_ref_aa(ts)
        out = []
        for forced in (0, 1):
            p = N.default_params(30., 10., top_k, 5)
            p.reserved[2] = forced
            h, c, st = ctx.search(p)
            out.append((h.tobytes(), c.tobytes(), st['candidates'], st['pairs'], st['tracebacks']))
        assert out[0] == out[1]
        oh, oc, ost = O.search(qs, ts, O.default_params(30., 10., top_k, 5))
        _cmp_hits(np.frombuffer(out[0][0], dtype=N.HIT_DTYPE), np.frombuffer(out[0][1], dtype=np.uint32), oh, oc)
        assert out[0][2] == ost['candidates']
    assert ost['candidates'] < 4096


def test_identical_pairs_settled_without_a_sweep(ctx):
    """the score pass settles identical pairs by comparison (identical_check): same table and statistics as with every pair swept
    (params.reserved2 bit 1) and as the oracle, on sets built to tempt it - exact duplicates, duplicates with one residue changed at either
    end or in the middle, sequences with X inside (swept) and at the end (the stop codon of a gene: settled), a protein that is a repeat of one short unit (the
    band over diagonal 0 next to bands over the shifted copies), prefixes of other proteins, equal lengths with different residues; and
    on the nucleotide configuration (N is not dominant)"""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    base = synth.make_proteins(60, length=(40, 420), seed=9, family=3, sub=0.15)
    extra = []
    for k in (0, 7, 19, 33):
        p = base[k].copy(); extra.append(p.copy())                                  # exact duplicate
        for at in (0, len(p) // 2, len(p) - 1):
            m = p.copy(); m[at] = (m[at] + 1 + rng.integers(0, 18)) % 20; extra.append(m)
        x = p.copy(); x[len(x) // 3] = 23; extra.append(x); extra.append(x.copy())  # X inside: a duplicate pair that must be swept
        extra.append(p[:len(p) - 7].copy()); extra.append(p[5:].copy())
    for k, m in ((2, 1), (13, 3), (21, 1)):                                         # the stop at the end of a gene: X behind the last residue
        y = np.concatenate([base[k], np.full(m, 23, dtype=np.uint8)]); extra.append(y); extra.append(y.copy())
    extra.append(np.concatenate([base[2], np.array([23, 0], dtype=np.uint8)]))          # ... and a residue behind the X: swept
    extra.append(np.full(40, 23, dtype=np.uint8))                                       # nothing but X
    unit = base[3][:23]
    extra.append(np.concatenate([unit] * 9)); extra.append(np.concatenate([unit] * 9))
    shuffled = base[11].copy(); rng.shuffle(shuffled); extra.append(shuffled)         # same length and composition, other order
    prots = base + extra
    keys = ('candidates', 'pairs', 'tracebacks', 'tracebacks_gapless', 'hits', 'cells', 'cells_trace')
    ctx.set_query_aa(prots); ctx.set_ref_aa(prots)
    out = {}
    for flag in (4, 2, 5, 3, 0):                                                    # bit 2: also in a search this small; 0 = production (too small: swept)
        p = N.default_params(30., 10., 25, 5)
        p.reserved2 = flag
        h, c, st = ctx.search(p)
        out[flag] = (h.tobytes(), c.tobytes(), tuple(st[k] for k in keys), st['candidates_settled'], st['cells_settled'], st['cells_swept'])
    assert out[4][:3] == out[2][:3] == out[5][:3] == out[3][:3] == out[0][:3]
    assert out[2][3] == 0 and out[2][4] == 0 and out[0][3] == 0 and out[4][3] == out[5][3] >= 60 and 0 < out[4][4] < dict(zip(keys, out[4][2]))['cells']
    assert out[4][5] < out[2][5]                                                    # fewer cells swept
    out[0] = out[4]
    oh, oc, ost = O.search(prots, prots, O.default_params(30., 10., 25, 5))
    _cmp_hits(np.frombuffer(out[0][0], dtype=N.HIT_DTYPE), np.frombuffer(out[0][1], dtype=np.uint32), oh, oc)
    assert ost['candidates'] == out[0][2][0] and ost['cells'] == dict(zip(keys, out[0][2]))['cells']
    # nucleotide configuration: exact 17-mers, +2 / -3; sequences with N are swept
    names, seqs = synth.make_genes(120, 0, seed=41)
    codes = [O.nt_codes(x) for x in seqs]
    codes += [codes[5].copy(), codes[9].copy()]
    withn = codes[12].copy(); withn[30:33] = 4; codes += [withn, withn.copy()]
    pn = N.nucleotide_params(70., 25.)
    ms = np.array([O.min_score(len(x), pn.dbsize, pn.max_evalue, pn.ka_lambda, pn.ka_k) for x in codes], dtype=np.int32)
    ctx.set_query_aa(codes); ctx.set_ref_aa(codes)
    res = {}
    for flag in (4, 2):
        pn.reserved2 = flag
        gh, gc, st = ctx.search(pn)
        res[flag] = (gh.tobytes(), gc.tobytes(), st['candidates'], st['cells'], st['candidates_settled'])
    res[0] = res[4]
    assert res[0][:4] == res[2][:4] and res[2][4] == 0 and res[0][4] >= 100
    pn.reserved2 = 0
    oh, oc, ost = O.search(codes, codes, O.params_from(pn), min_scores=ms)
    _cmp_hits(np.frombuffer(res[0][0], dtype=N.HIT_DTYPE), np.frombuffer(res[0][1], dtype=np.uint32), oh, oc)


def test_k1_inside_the_search_equals_translate_in_front(ctx):
    """pep_invalidate_translation: K1 runs inside the next search (both sides queued at once, the reference side's summary taken late, the
    host tables built from descriptors that stay on the device) - same hit table, same statistics, same meta records and proteins as
    translate(force=True) in front of the search; also when only one side changes, and several times in a row"""
    from peppan_amd import _native as N, synth
    names, seqs = synth.make_genes(700, 0, seed=5)
    keys = ('candidates', 'pairs', 'tracebacks', 'hits', 'cells', 'query_residues', 'target_residues', 'query_seeds', 'target_seeds', 'seed_hits')
    p = N.default_params(45., 25., 10, 5)
    ctx.set_query_nt(seqs[:300], 11); ctx.set_ref_nt(seqs, 6, 11)
    ctx.translate(force=True)
    h0, c0, s0 = ctx.search(p)
    qm0, tm0 = ctx.query_meta(), ctx.target_meta()
    qa0, ta0 = ctx.query_aa(), ctx.target_aa()
    for rep in range(3):
        ctx.invalidate_translation()
        h, c, st = ctx.search(p, copy=bool(rep % 2))
        assert np.array(h).tobytes() == h0.tobytes() and np.array(c).tobytes() == c0.tobytes() and all(st[k] == s0[k] for k in keys), rep
    assert ctx.query_meta().tobytes() == qm0.tobytes() and ctx.target_meta().tobytes() == tm0.tobytes()
    qa, ta = ctx.query_aa(), ctx.target_aa()
    assert all(np.array_equal(a, b) for a, b in zip(qa + ta, qa0 + ta0))
    # only the query side is new: the reference side is not translated again, the result is that of a fresh context
    ctx.set_query_nt(seqs[100:500], 11)
    h1, c1, s1 = ctx.search(p)
    with N.Context(0) as other:
        other.set_query_nt(seqs[100:500], 11); other.set_ref_nt(seqs, 6, 11)
        other.translate()
        h2, c2, s2 = other.search(p)
        tm2 = other.target_meta()
    assert h1.tobytes() == h2.tobytes() and c1.tobytes() == c2.tobytes() and all(s1[k] == s2[k] for k in keys) and len(h1) > 400
    assert ctx.target_meta().tobytes() == tm2.tobytes()
    # an empty reference set / empty query set inside the search
    ctx.set_ref_nt([], 6, 11)
    h3, c3, s3 = ctx.search(p)
    assert len(h3) == 0
    ctx.set_query_nt([], 11); ctx.set_ref_nt(seqs[:50], 6, 11)
    h4, c4, s4 = ctx.search(p)
    assert len(h4) == 0 and len(ctx.target_meta()) >= 300


@pytest.mark.gpu
def test_blast_hsp_mode_2_equals_the_oracle(ctx):
    """hsp_mode 2 (BLAST's way with the HSPs of a subject; the definition and its made-up cases: tests/test_hsp_mode.py): the kernels - cull_hsps,
    subject_best, topk's subject ranking - against the oracle, bit for bit.  Genes with tandem repeats and second copies through the nucleotide tool's
    own layout (forward strands, then reverse complements: the two strands of a sequence are ONE subject), hit lists short enough to cut (top_k 1000 /
    3 / 1), a contig with several copies of a gene as a raw target, and mode 1 on the same data as the superset."""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    names, seqs = synth.make_genes(150, 0, seed=9)
    texts = [s.decode() for s in seqs]
    for i in range(0, 150, 3):
        t = texts[i]
        a, b = sorted(rng.integers(50, len(t) - 50, size=2).tolist())
        rep = t[a:a + 90]
        texts[i] = t[:a + 90] + rep + t[a + 90:b] + rep[:70] + t[b:]
    texts[7] = texts[6]                                    # identical genes: ties in score, decided by position
    codes = [O.nt_codes(t) for t in texts]
    rc = [(3 - c[::-1]).astype(np.uint8) for c in codes]
    ctx.set_query_nt(texts)
    ctx.set_ref_nt(texts, 6, 11)
    subjects = list(range(len(codes))) * 2
    sizes = {}
    for top_k in (1000, 3, 1):
        for mode in (2, 1):
            p = N.nucleotide_params(60., 10., top_k=top_k, hsp_mode=mode)
            ctx.use_nt_as_residues(2)
            gh, gc, st = ctx.search(p)
            ms = np.array([O.min_score(len(c), p.dbsize, p.max_evalue, p.ka_lambda, p.ka_k) for c in codes], dtype=np.int32)
            oh, oc, ost = O.search(codes, codes + rc, O.params_from(p), min_scores=ms, subjects=subjects if mode == 2 else None)
            _cmp_hits(gh, gc, oh, oc)
            sizes[(top_k, mode)] = len(gh)
    print(sizes)
    assert sizes[(1000, 2)] < sizes[(1000, 1)]             # nested HSPs went
    assert sizes[(1, 1)] == len(codes) <= sizes[(1, 2)]    # one SUBJECT per query: at least its best alignment
    assert sizes[(3, 2)] < sizes[(1000, 2)]                # and the list does cut
    # raw targets (every target its own subject): a contig with two copies of gene 0 and one of gene 4 per strand
    sp = lambda n: rng.integers(0, 4, n).astype(np.uint8)
    def mutated(c, rate):
        c = c.copy(); m = rng.random(len(c)) < rate; c[m] = rng.integers(0, 4, int(m.sum())); return c
    comp = lambda c: (3 - c[::-1]).astype(np.uint8)
    contig = np.concatenate([sp(700), codes[0], sp(333), mutated(codes[0], 0.04), sp(90), comp(codes[4]), sp(2000), mutated(codes[4], 0.08), sp(50)])
    targets = [contig, comp(contig)] + codes[:10]
    ctx.set_query_aa(codes[:20])
    ctx.set_ref_aa(targets)
    p = N.nucleotide_params(70., 25., hsp_mode=2)
    gh, gc, st = ctx.search(p)
    ms = np.array([O.min_score(len(c), p.dbsize, p.max_evalue, p.ka_lambda, p.ka_k) for c in codes[:20]], dtype=np.int32)
    oh, oc, ost = O.search(codes[:20], targets, O.params_from(p), min_scores=ms)
    _cmp_hits(gh, gc, oh, oc)
    assert ((gh['q'] == 0) & (gh['t'] == 0) & (gh['q_end'] - gh['q_start'] > 0.9 * len(codes[0]))).sum() == 2       # both copies of gene 0 stay: neither lies inside the other


@pytest.mark.gpu
def test_rescoring_inside_the_search_equals_k7_behind_it(tmp_path, monkeypatch):
    """-s 1 (reScore mode 1, uberBlast.py:397-415).  The product has the search count K7's identical nucleotide columns for its own hits from the table on the
    device (pep_set_nt_match) and pep_table_from_hits make identity and score from them; PEPPAN_NT_MATCH_IN_SEARCH=0 is the chain as it was - table built, uploaded
    again, K7 (pep_rescore_nt), numpy arithmetic.  Same tables, cell for cell: the hot call's shape (genes against themselves and their mutated relatives, both
    tools side by side), and the mapping's (genes against genomes in one batch: both strands, frames cut into chunks at stop codons - chunk offsets -, contigs with
    N runs and lower-case stretches - K7's 'other' code -, gaps of one to several codons)"""
    import io, contextlib
    from peppan_amd import uberBlast as UB, synth, configure
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(41)
    names, seqs = synth.make_genes(120, 0, seed=23, family=4)
    seqs = [bytearray(s) for s in seqs]
    for s in seqs[::7]:                                      # N runs and lower-case stretches inside genes
        a = int(rng.integers(30, len(s) - 60))
        s[a:a + int(rng.integers(1, 9))] = b'N' * 8
        b = int(rng.integers(30, len(s) - 60))
        s[b:b + 25] = bytes(s[b:b + 25]).lower()
    seqs = [bytes(s) for s in seqs]
    _write_fasta('genes.fa', [str(i) for i in range(len(seqs))], seqs)

    def mutated(s):                                          # substitutions, a deleted codon or two, an inserted base triple
        s = bytearray(s.upper())
        for k in rng.integers(0, len(s), max(1, len(s) // 25)):
            s[k] = b'ACGT'[int(rng.integers(0, 4))]
        a = int(rng.integers(60, len(s) - 60))
        del s[a:a + 3 * int(rng.integers(1, 3))]
        b = int(rng.integers(60, len(s) - 60))
        s[b:b] = b'GCA' * int(rng.integers(1, 3))
        return bytes(s)
    spacer = lambda: bytes(rng.choice(list(b'ACGTN'), int(rng.integers(40, 300)), p=[.24, .24, .24, .24, .04]).tolist())
    genomes = []
    for g in range(3):
        contigs, cur = [], spacer()
        for k in rng.permutation(len(seqs))[:60]:
            s = mutated(seqs[k]) if rng.random() < 0.6 else seqs[k]
            if rng.random() < 0.5:
                s = configure.rc(s.decode()).encode()
            cur += s + spacer()
            if rng.random() < 0.1:
                contigs.append(cur); cur = spacer()
        contigs.append(cur)
        _write_fasta('genome%d.fa' % g, ['%d:c%d' % (g, i) for i in range(len(contigs))], contigs)
        genomes.append('genome%d.fa' % g)
    hot = '-r genes.fa -q genes.fa --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 2 --min_ratio 0.25 -e 3,3 -p --gtable 11'.split()
    mapping = '-q genes.fa -f -m -O --blastn --diamond --min_id 0.4 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11'.split()
    out = {}
    with contextlib.redirect_stderr(io.StringIO()):
        for mode in ('1', '0'):
            monkeypatch.setenv('PEPPAN_NT_MATCH_IN_SEARCH', mode)
            out[mode] = (UB.uberBlast(hot), UB.uberBlastBatch(genomes, mapping), UB.uberBlast(['-r', genomes[0]] + mapping))
    (h1, b1, s1), (h0, b0, s0) = out['1'], out['0']
    assert h1.shape[0] > 300 and h1.tolist() == h0.tolist()
    assert len({r[2] for r in h1.tolist()}) > 30                       # (identities of many values: the relatives are mutated)
    for (t1, o1), (t0, o0) in zip(b1 + [s1], b0 + [s0]):
        assert t1.shape[0] > 50 and t1.tolist() == t0.tolist() and o1.tolist() == o0.tolist()
    assert any(r[8] > r[9] for t, _ in b1 for r in t.tolist()) and any(r[8] < r[9] for t, _ in b1 for r in t.tolist())      # both strands
    assert any('I' in r[14] or 'D' in r[14] for t, _ in b1 for r in t.tolist())                                             # gapped alignments among them
