"""hsp_mode 2 - BLAST's way with the HSPs of a subject (the blastn call of uberBlast.py:294 with `-num_alignments 1000`) - as the oracle defines it
(oracle/align_oracle.c, align_group / oracle_search): made-up cases with known answers, and the properties that tie it to hsp_mode 1.  No blastn binary
exists here: the rule is NCBI's published behaviour restated (common start / end points purged, HSPs inside a better one dropped, hit list counted in
subjects), parity of the kernels with this definition is in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from oracle import oracle as O
from peppan_amd import _native as N


def _params(mode, top_k=1000, min_id=0., min_cov=0.):
    p = O.params_from(N.nucleotide_params(min_id, min_cov, top_k=top_k, hsp_mode=2 if mode == 2 else 1))
    p.hsp_mode = mode
    return p


def _boxes(hits):
    return sorted((int(h['q']), int(h['t']), int(h['q_start']), int(h['q_end']), int(h['t_start']), int(h['t_end']), int(h['score'])) for h in hits)


def test_an_hsp_inside_a_better_one_of_the_same_subject_strand_is_dropped():
    rng = np.random.default_rng(11)
    rnd = lambda n: rng.integers(0, 4, size=n).astype(np.uint8)
    A, R, B = rnd(150), rnd(120), rnd(150)
    Q = np.concatenate([A, R, R, B])                      # a gene with a tandem repeat against itself: the repeat's copies align to each other 120 diagonals off
    h1 = _boxes(O.search([Q], [Q.copy(), rnd(500)], _params(1), max_evalue=1e-2)[0])
    h2 = _boxes(O.search([Q], [Q.copy(), rnd(500)], _params(2), max_evalue=1e-2)[0])
    assert h1 == [(0, 0, 1, 540, 1, 540, 1080), (0, 0, 151, 270, 271, 390, 240), (0, 0, 271, 390, 151, 270, 240)]
    assert h2 == [(0, 0, 1, 540, 1, 540, 1080)]           # both off-diagonal HSPs lie inside the full-length one's ranges
    # a SECOND copy of the gene further along the subject is not inside the first: both stay in either mode
    T = np.concatenate([rnd(40), Q, rnd(300), Q[:400], rnd(40)])
    for mode in (1, 2):
        got = _boxes(O.search([Q], [T], _params(mode), max_evalue=1e-2)[0])
        long_ones = [b for b in got if b[3] - b[2] >= 390]
        assert len(long_ones) == 2 and {b[4] for b in long_ones} == {41, 881}


def test_the_hit_list_counts_subjects_and_keeps_every_hsp_of_a_kept_subject():
    rng = np.random.default_rng(12)
    rnd = lambda n: rng.integers(0, 4, size=n).astype(np.uint8)
    Q = rnd(540)
    T0, T1, T2 = Q.copy(), np.concatenate([rnd(50), Q[100:400], rnd(50)]), np.concatenate([rnd(30), Q[0:450], rnd(30)])
    run = lambda mode, k, subj: sorted(int(h['t']) for h in O.search([Q], [T0, T1, T2], _params(mode, top_k=k), max_evalue=1e-2, subjects=subj)[0])
    assert run(1, 1, None) == [0] and run(1, 2, None) == [0, 2]                 # per alignment: the best, the two best
    assert run(2, 1, [0, 0, 1]) == [0, 1]                                       # targets 0 and 1 are one subject: its two alignments stay, subject 1 goes
    assert run(2, 2, [0, 0, 1]) == [0, 1, 2]
    assert run(2, 1, None) == [0] and run(2, 2, [0, 1, 2]) == [0, 2]            # every target a subject of its own: as per alignment when no subject has two
    assert run(2, 1, [1, 0, 0]) == [0]                                          # the best subject is the one with the best alignment, whatever its number


def test_mode_2_keeps_a_subset_of_mode_1_and_every_dropped_hsp_has_a_reason():
    from peppan_amd import synth
    rng = np.random.default_rng(5)
    names, seqs = synth.make_genes(60, 0, seed=9)
    codes = [O.nt_codes(s.decode()) for s in seqs]
    for i in range(0, 60, 3):                               # tandem repeats and a second copy of a stretch inside every third gene
        c = codes[i]
        a, b = sorted(rng.integers(50, len(c) - 50, size=2).tolist())
        rep = c[a:a + 90]
        codes[i] = np.concatenate([c[:a + 90], rep, c[a + 90:b], rep[:70], c[b:]])
    rc = [(3 - c[::-1]).astype(np.uint8) for c in codes]
    targets, subjects = codes + rc, list(range(len(codes))) * 2
    h1, c1, _ = O.search(codes, targets, _params(1, min_id=60., min_cov=10.), max_evalue=1e-2)
    h2, c2, _ = O.search(codes, targets, _params(2, min_id=60., min_cov=10.), max_evalue=1e-2, subjects=subjects)
    b1, b2 = _boxes(h1), _boxes(h2)
    assert set(b2) <= set(b1) and 0 < len(b1) - len(b2) < len(b1) // 2
    kept = {}
    for b in b2:
        kept.setdefault(b[:2], []).append(b)
    for b in set(b1) - set(b2):                             # (top_k 1000 subjects is never reached here: every loss is a culling)
        assert any(k[6] >= b[6] and ((k[2], k[4]) == (b[2], b[4]) or (k[3], k[5]) == (b[3], b[5]) or (k[2] <= b[2] and b[3] <= k[3] and k[4] <= b[4] and b[5] <= k[5]))
                   for k in kept.get(b[:2], [])), b
    # the CIGARs of what stays are those of mode 1
    runs1 = {(int(h['q']), int(h['t']), int(h['q_start']), int(h['t_start']), int(h['q_end'])): c1[int(h['cigar_off']):int(h['cigar_off']) + int(h['cigar_runs'])].tolist() for h in h1}
    for h in h2:
        assert runs1[(int(h['q']), int(h['t']), int(h['q_start']), int(h['t_start']), int(h['q_end']))] == c2[int(h['cigar_off']):int(h['cigar_off']) + int(h['cigar_runs'])].tolist()


def test_the_drop_in_passes_the_mode_on(monkeypatch, tmp_path):
    """RunBlast.blast_hsp_mode (PEPPAN_BLAST_HSP_MODE) reaches the nucleotide tool's search; the default stays 1"""
    from oracle_context import OracleContext
    from peppan_amd import uberBlast as UB
    rng = np.random.default_rng(11)
    rnd = lambda n: ''.join('ACGT'[i] for i in rng.integers(0, 4, size=n))
    A, R, B = rnd(150), rnd(120), rnd(150)
    fa = tmp_path / 'g.fa'
    fa.write_text('>g1\n%s\n>g2\n%s\n' % (A + R + R + B, rnd(400)))
    ctx = OracleContext()
    monkeypatch.setattr(UB, 'get_context', lambda device=None: ctx)
    argv = ('-r %s -q %s --blastn --min_id 0.5 --min_cov 40 --min_ratio 0.05 -t 1 -p' % (fa, fa)).split()
    assert UB.RunBlast().blast_hsp_mode == 1
    rows1 = UB.uberBlast(argv)
    monkeypatch.setenv('PEPPAN_BLAST_HSP_MODE', '2')
    rows2 = UB.uberBlast(argv)
    own = lambda rows: sorted((int(r[6]), int(r[7]), int(r[8]), int(r[9])) for r in rows if r[0] == 'g1' and r[1] == 'g1')
    assert own(rows1) == [(1, 540, 1, 540), (151, 270, 271, 390), (271, 390, 151, 270)] and own(rows2) == [(1, 540, 1, 540)]
    with pytest.raises(ValueError):
        N.nucleotide_params(hsp_mode=0)
