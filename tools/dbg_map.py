import os, sys, tempfile, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from peppan_amd import mapbsn, synth, uberBlast as UB
os.chdir(tempfile.mkdtemp())
names, seqs = synth.make_genes(120, 0, seed=23, family=4)
with open('cl', 'w') as f:
    for i, s in enumerate(seqs): f.write('>%d\n%s\n' % (i, s.decode()))
worlds = synth.make_genomes(seqs, 5, seed=77)
gname, contig, ann = worlds[0]
with open('g', 'w') as f: f.write('>5000\n%s\n' % contig.decode())
print([a for a in ann if a[0] in (96, 97, 98, 99)])
for flags in ('--blastn --diamond -s 1', '-f --blastn --diamond -s 1', '-f -m --blastn --diamond -s 1'):
    tab = UB.uberBlast(('-r g -q cl %s --min_id 0.55 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -e 0,3 --gtable 11' % flags).split())
    print(flags, len(tab))
    for r in tab:
        if r[0] in ('96', '97'): print('   ', [x for i, x in enumerate(r) if i != 14][:17])
