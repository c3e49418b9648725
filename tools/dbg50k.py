import sys, os, io, contextlib, tempfile
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import mapbsn, synth, uberBlast as UB
os.chdir(tempfile.mkdtemp())
names, seqs = synth.make_genes(50000, 0, seed=355)
with open('m.clust.exemplar', 'w') as f:
    for i, s in enumerate(seqs): f.write('>%d\n%s\n' % (i, s.decode()))
worlds = synth.make_genomes(seqs, 5, seed=355, presence=synth.PAN_GENOME_PRESENCE)
g = 4
gname, contig, ann = worlds[g]
params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
              match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
open('g.fa', 'w').write('>%d\n%s\n' % (100000 + g, contig.decode()))
k = 39428
loc = [a for a in ann if a[0] == k][0]
print('planted', loc, 'gene len', len(seqs[k]), 'family members', [len(seqs[x]) for x in range(k - k % 4, k - k % 4 + 4)])
for flags in ('', '-f', '-f -m'):
    argv = ['-r', 'g.fa', '-q', 'm.clust.exemplar'] + flags.split() + '--blastn --diamond --min_id 0.55 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11'.split()
    with contextlib.redirect_stderr(io.StringIO()):
        tab = UB.uberBlast(argv)
    rows = [r for r in tab if (min(r[8], r[9]) <= loc[2] and max(r[8], r[9]) >= loc[1])]
    print('flags', flags, 'rows at the locus:')
    for r in rows: print('   ', [r[0], r[1], r[2], r[3], r[6], r[7], r[8], r[9], r[11], r[12]], r[16] if len(r) > 16 else '')
