"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference's Python."""
import os

import numpy as np
import pytest
from conftest import load_golden
from oracle import oracle as O


def test_blosum62_matches_reference_table():
    g = load_golden('g01_tables.json')
    flat = np.array(g['blosum62'])                      # index (ord(r)-65)*32 + ord(q)-65, configure.py:48
    ref = np.zeros(1024, dtype=int); ref[:len(flat)] = flat; ref = ref.reshape(32, 32)
    p = O.default_params()
    sub = np.array(list(p.sub), dtype=int).reshape(32, 32)
    letters = 'ARNDCQEGHILKMFPSTWYVBZX'
    for a in letters:
        for b in letters:
            assert sub[ord(a) - 65, ord(b) - 65] == ref[ord(a) - 65, ord(b) - 65], (a, b)


def _frames(spec):
    m = {'F': [1, 2, 3], 'R': [4, 5, 6], '7': [1, 2, 3, 4, 5, 6]}.get(str(spec).upper())
    return m if m else [int(f) for f in str(spec).split(',')]


def test_transeq_golden():
    g = load_golden('g01_transeq.json')
    n = 0
    for case in g['cases']:
        if case.get('markStarts') or case['table'] == 'starts':
            continue
        for name, nt in g['seqs'].items():
            got = O.translate_frames(nt, _frames(case['frame']), case['table'])
            exp = case['out'][name] if isinstance(case['out'], dict) else dict(case['out'])[name]
            assert got == exp, (case['frame'], case['table'], name)
            n += 1
    assert n > 100


@pytest.mark.parametrize('frames', ['7', 'F'])
def test_rundiamond_fasta_golden(frames):
    g = load_golden('g02_rundiamond.json')
    q_txt, r_txt = O.diamond_fasta(g['query'], g['ref'], frames, g['table_id'])
    assert q_txt == g['out'][frames]['qryAA']
    assert r_txt == g['out'][frames]['refAA']
    # chunking really happened (a frame longer than 1000 aa was cut)
    assert any(':' in l and not l.endswith(':0') for t in r_txt for l in t.split('\n') if l.startswith('>'))


def test_rescore_counts_golden():
    g = load_golden('g05_rescore.json')
    case = [c for c in g['cases'] if c['mode'] == 1 and c['table_id'] == 11][0]
    exp = {r[15]: r for r in case['rows']}
    q = {k: O.nt_encode_rescore(v.upper()) for k, v in g['query'].items()}
    r = {k: O.nt_encode_rescore(v.upper()) for k, v in g['ref'].items()}
    opc = {'M': 0, 'I': 1, 'D': 2}
    checked = 0
    for row in g['table']:
        cig = np.array([(n << 2) | opc[o] for n, o in row[14]], dtype=np.uint32)
        c = O.rescore_counts(q[row[0]], r[row[1]], row[6], row[8], row[9], cig)
        nmatch, nmis, ngap, bgap, mgap = [int(x) for x in c]
        iden = np.round(float(nmatch) / (nmatch + nmis + bgap - mgap), 3)
        score = np.round(float(nmatch * 3 - nmis - ngap * 5 - bgap), 3)
        if row[15] in exp:
            assert iden == exp[row[15]][2] and score == exp[row[15]][11], row[:2]
            checked += 1
        else:
            assert iden < case['min_id']
    assert checked == len(exp) and checked > 20


def test_components_matches_reference_partition():
    g = load_golden('g11_groups.json')
    for case in g['cases']:
        clu, bsn = np.array(case['clu']), np.array(case['bsn'])
        e = np.vstack([clu[:, :2], bsn[bsn[:, 2] > 0][:, :2]])
        lab = O.components(60, e[:, 0], e[:, 1])
        ref_sets = {frozenset(m) for _, m in case['groups']}
        got = {}
        for i, l in enumerate(lab):
            got.setdefault(int(l), set()).add(i)
        got_sets = {frozenset(s) for s in got.values() if len(s) > 1}
        assert got_sets == ref_sets


def test_oracle_search_self_consistency():
    rng = np.random.default_rng(5)
    aa = 'ARNDCQEGHILKMFPSTWYV'
    prots = []
    for _ in range(6):
        root = rng.integers(0, 20, 200)
        prots.append(root)
        m = root.copy()
        pos = rng.random(200) < 0.15
        m[pos] = rng.integers(0, 20, pos.sum())
        m = np.concatenate([m[:80], m[84:]])       # 4-residue deletion
        prots.append(m)
    seqs = [O.aa_codes(''.join(aa[i] for i in p)) for p in prots]
    hits, cig, st = O.search(seqs, seqs, O.default_params(top_k=10, n_splits=5))
    pairs = {(int(h['q']), int(h['t'])) for h in hits}
    for i in range(0, 12, 2):
        assert (i, i) in pairs and (i, i + 1) in pairs and (i + 1, i) in pairs
    for h in hits:
        runs = cig[h['cigar_off']:h['cigar_off'] + h['cigar_runs']]
        ops, lens = runs & 3, runs >> 2
        assert lens[ops != 2].sum() == h['q_end'] - h['q_start'] + 1
        assert lens[ops != 1].sum() == h['t_end'] - h['t_start'] + 1
        assert lens.sum() == h['aln_len'] and ops[0] == 0 and ops[-1] == 0
        if h['q'] == h['t']:
            assert len(runs) == 1 and h['n_ident'] == h['aln_len'] == len(seqs[h['q']])
    # the deletion shows up as one 4-long I run (query residues absent from the target)
    h = [h for h in hits if h['q'] == 0 and h['t'] == 1][0]
    runs = cig[h['cigar_off']:h['cigar_off'] + h['cigar_runs']]
    assert list(runs & 3) == [0, 1, 0] and (runs >> 2)[1] == 4


def test_linclust_gapped_verification():
    """K9 restatement: an allele that differs from its centre by an indel fails every ungapped diagonal and is absorbed by the
    gapped verification (identity over alignment columns, coverage of both sequences by the aligned span)"""
    rng = np.random.default_rng(99)
    centre = rng.integers(0, 4, 900).astype(np.uint8)

    def variant(n_sub, cut=None, ins=None):
        v = centre.copy()
        for p in rng.choice(len(v), n_sub, replace=False):
            v[p] = (v[p] + 1) % 4
        if cut:
            v = np.concatenate([v[:cut[0]], v[cut[0] + cut[1]:]])
        if ins:
            v = np.concatenate([v[:ins[0]], rng.integers(0, 4, ins[1]).astype(np.uint8), v[ins[0]:]])
        return v
    seqs = [centre, variant(10, cut=(450, 3)), variant(12, ins=(300, 6)), variant(8), variant(200, cut=(500, 9)),
            rng.integers(0, 4, 880).astype(np.uint8), variant(5, cut=(100, 360))]
    rep, st = O.linclust(seqs, 0.95, 0.9)
    # the longest sequence (index 2, carries a 6-nt insertion) is the centre: every other allele needs gaps to reach it
    assert rep.tolist()[:4] == [2, 2, 2, 2]
    assert rep[4] == 4 and rep[5] == 5                  # too diverged / unrelated
    assert rep[6] == 6                                  # the part after its 360-nt deletion covers 49 % of the centre
    rep2, _ = O.linclust(seqs, 0.95, 0.4)
    assert rep2[6] == 2                                 # ... which is enough at coverage 0.4
    rep3, st3 = O.linclust([centre, seqs[3]], 0.95, 0.9)
    assert rep3.tolist() == [0, 0] and st3['accepted'] == 1          # substitutions only: accepted on the diagonal, no gaps needed


def test_rule_5a_changes_no_reported_alignment():
    """Rule 5a (the gapless shortcut) came into the oracle together with its GPU counterpart.  With the oracle's switch off every
    alignment is traced as before the rule existed: the tables must be the same - protein families with indels and duplicates, the nucleotide
    configuration on both strands, and real genes (golden G16) - while the shortcut settles a good part of the pairs"""
    import gzip
    from conftest import GOLDEN
    from peppan_amd import synth

    def both(q, t, p):
        out = []
        for on in (True, False):
            O.set_rule5a(on)
            O.trace_counts(reset=True)
            h, c, st = O.search(q, t, p)
            out.append((h, c, O.trace_counts()))
        O.set_rule5a(True)
        return out
    fields = ('q', 't', 'q_start', 'q_end', 't_start', 't_end', 'score', 'nm', 'n_ident', 'aln_len', 'cigar_runs', 'bin')
    cases = []
    for seed in (0, 1):
        prots = synth.make_proteins(300, length=(60, 400), seed=seed, family=4, sub=0.25, indel=0.7)
        cases.append((prots + prots[:20], prots, O.default_params(30., 20., 10, 5)))
    names, genes = synth.make_genes(120, 0, seed=5)
    from peppan_amd import _native as N
    nt = [O.nt_codes(s.decode()) for s in genes]
    cases.append((nt, nt + [(3 - x[::-1]).astype(np.uint8) for x in nt], O.params_from(N.nucleotide_params(70., 20.))))
    with gzip.open(os.path.join(GOLDEN, 'g16_real_genes.fa.gz'), 'rt') as f:
        real = [s.split('\n', 1)[1].replace('\n', '') for s in f.read().split('>')[1:]][:250]
    q_aa = [O.aa_codes(O.query_frame(s, 11)[1]) for s in real]
    cases.append((q_aa, q_aa, O.default_params(45., 25., 10, 5)))
    settled = 0
    for q, t, p in cases:
        (h1, c1, k1), (h0, c0, k0) = both(q, t, p)
        assert k0['gapless'] == 0 and k1['traced'] == k0['traced']
        assert len(h1) == len(h0) and all(np.array_equal(h1[f], h0[f]) for f in fields) and np.array_equal(c1, c0)
        settled += k1['gapless']
    assert settled > 800
