"""Reference sequences longer than the engine's per-sequence limit (PEP_MAX_SEQ_LEN, 8.39 Mbp) in the nucleotide tool: runBlast searches
them as overlapping windows and maps the hits back (a long chromosome used to cost the whole blastn table).  CPU test: the
oracle-backed context, with the limit and the window size patched down so that a 16 kb contig needs eight windows."""
import contextlib
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_tiled_nucleotide_search_equals_untiled(tmp_path, monkeypatch):
    from oracle_context import OracleContext
    from peppan_amd import uberBlast as UB, _native as N, synth
    names, seqs = synth.make_genes(24, 0, seed=12)
    genes = [s.decode() for s in seqs if len(s) <= 900][:12]
    rng = np.random.default_rng(3)
    comp = str.maketrans('ACGT', 'TGCA')
    contig, placed = [], 0
    for k in range(3):                                   # every gene three times, mutated, on alternating strands, with spacers
        for g in genes:
            s = np.frombuffer(g.encode(), dtype=np.uint8).copy()
            m = rng.random(len(s)) < 0.03
            s[m] = np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
            s = s.tobytes().decode()
            contig.append(''.join('ACGT'[x] for x in rng.integers(0, 4, int(rng.integers(20, 200)))))
            contig.append(s if (placed % 2 == 0) else s.translate(comp)[::-1])
            placed += 1
    contig = ''.join(contig)
    assert len(contig) > 14000
    with open(tmp_path / 'q.fa', 'w') as f:
        for i, g in enumerate(genes):
            f.write('>%d\n%s\n' % (i, g))
    with open(tmp_path / 'r.fa', 'w') as f:
        f.write('>chr\n%s\n>small\n%s\n' % (contig, contig[5000:7000]))
    octx = OracleContext()
    monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
    argv = ('-r %s -q %s --blastn -s 1 --min_id 0.6 --min_cov 50 --min_ratio 0.2 -e 0,3 -f -m -O' % (tmp_path / 'r.fa', tmp_path / 'q.fa')).split()
    with contextlib.redirect_stderr(io.StringIO()):
        whole, whole_ovl = UB.uberBlast(argv)
        monkeypatch.setattr(N, 'MAX_SEQ_LEN', 4400)
        monkeypatch.setattr(UB, '_TILE_HOME', 2048)
        UB._FASTA_CACHE.clear()
        tiled, tiled_ovl = UB.uberBlast(argv)
    assert whole.shape[0] >= 3 * len(genes) and any(r[8] > r[9] for r in whole)          # both strands, every copy
    assert tiled.tolist() == whole.tolist() and tiled_ovl.tolist() == whole_ovl.tolist()
    # references handed over in memory as ASCII bytes (what the mapping workers pass) take the same road: text and bytes encode alike
    codes, off = UB._encode_nt([contig[:700], contig[700:900].encode(), ''])
    both, off2 = UB._encode_nt([contig[:700], contig[700:900], ''])
    assert codes.tolist() == both.tolist() and off.tolist() == off2.tolist() == [0, 700, 900, 900]
    assert UB._reference_seqs([('chr x', contig[:50].lower().encode()), (7, b'AC GT')]) == {'chr': contig[:50].encode(), '7': b'ACGT'}
    # queries too long for the windows are a loud error, not a silently empty table
    monkeypatch.setattr(N, 'MAX_SEQ_LEN', 3000)
    import pytest
    with pytest.raises(N.PepError):
        with contextlib.redirect_stderr(io.StringIO()):
            UB.uberBlast(argv)
