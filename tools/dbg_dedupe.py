import sys, io, contextlib, os, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from peppan_amd import _native as N, synth, configure
rng = np.random.default_rng(12)
names, seqs = synth.make_genes(80, 0, seed=19, family=4)
spacer = lambda: bytes(rng.choice(list(b'ACGT'), int(rng.integers(40, 400))).tolist())
genomes = []
for g in range(4):
    contigs, cur = [], spacer()
    for k in rng.permutation(len(seqs))[:45]:
        s = seqs[k]
        if rng.random() < 0.5: s = configure.rc(s.decode()).encode()
        cur += s + spacer()
        if rng.random() < 0.12: contigs.append(cur); cur = spacer()
    contigs.append(cur + seqs[3] + spacer() + seqs[3] + spacer())
    genomes.append(contigs)
ctx = N.Context(0)
from oracle import oracle as O
def nt_search(contig_sets, dbg):
    p = N.nucleotide_params(40., 25.)
    p.reserved[0] = dbg
    q = [O.nt_codes(s.decode()) for s in seqs]
    t = []
    for cs in contig_sets:
        for c in cs:
            t.append(O.nt_codes(c.decode())); t.append(O.nt_codes(configure.rc(c.decode())))
    ctx.set_query_aa(q); ctx.set_ref_aa(t)
    h, c, st = ctx.search(p)
    return h, st
for sets in ([genomes[0]], [genomes[1]], genomes[:2], genomes[2:], genomes):
    a, sa = nt_search(sets, 0)
    b, sb = nt_search(sets, 9)
    print(len(sets), len(a), len(b), a.tobytes() == b.tobytes(), sa['candidates'], sb['candidates'], sa['seed_hits'], sb['seed_hits'])
