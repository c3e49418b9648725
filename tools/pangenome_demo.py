#!/usr/bin/env python3
"""End-to-end walk through the hot path with the reference-shaped entry points on synthetic data:
   writeGenes -> iterClust (K9) -> get_similar_pairs (K1-K8 both tools, K7) -> get_gene_group / GPU labels (K10)
   -> uberBlastBatch mapping of the exemplars against every genome (K1-K8, K7, K11, -f -m -O).
   python tools/pangenome_demo.py [n_genes=2000] [n_genomes=20]"""
import contextlib
import hashlib
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from peppan_amd import synth, pipeline as PL, uberBlast as UB

n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_genomes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
work = tempfile.mkdtemp(prefix='peppan_demo_')
os.chdir(work)
names, seqs = synth.make_genes(n_genes, 0, seed=355)
genomes = synth.make_genomes(seqs, n_genomes, seed=355)
genes = {i: ['f', '', 0, 0, '+', int(hashlib.sha1(s).hexdigest(), 16), s.decode()] for i, s in enumerate(seqs)}
prio = {i: [0, -len(s), genes[i][5]] for i, s in enumerate(seqs)}
params = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=1, match_frag_prop=0.25, gtable=11, clust_identity=0.9,
              clust_match_prop=0.8, incompleteCDS='', match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8,
              match_prop2=0.4)
T = {}


def timed(name, fn, *a, **k):
    t0 = time.perf_counter()
    with contextlib.redirect_stderr(io.StringIO()):
        r = fn(*a, **k)
    T[name] = time.perf_counter() - t0
    return r


fn, groups = timed('writeGenes', PL.writeGenes, 'p.genes', genes, prio)
ex = timed('iterClust (11 steps, K9)', PL.iterClust, 'p', fn, groups, dict(identity=0.9, coverage=0.8, n_thread=1, translate=False))
pairs = timed('get_similar_pairs (search + decision pass)', PL.get_similar_pairs, ex, prio, dict(params, clust=ex))
np.save('p.self_bsn.npy', pairs)
grp = timed('get_gene_group (host dict)', PL.get_gene_group, ex, 'p.self_bsn.npy')
lab = timed('gene_group_labels (K10)', PL.gene_group_labels, ex, 'p.self_bsn.npy', n_genes)
files = []
for name, contig, ann in genomes:
    with open(name + '.fa', 'w') as f:
        f.write('>%s:c1\n%s\n' % (name, contig.decode()))
    files.append(name + '.fa')
flags = '-q %s -f -m -O --blastn --diamond --min_id 0.4 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11' % ex
res = timed('uberBlastBatch (%d genomes, one search per tool)' % n_genomes, UB.uberBlastBatch, files, flags.split())
n_ex = open(ex).read().count('>')
print('genes %d -> exemplars %d; ortholog/conflict pairs %d; gene groups %d (labels: %d clusters); mapping rows per genome: %.0f'
      % (n_genes, n_ex, len(pairs), len(grp), len(np.unique(lab)), np.mean([r[0].shape[0] for r in res])))
for k, v in T.items():
    print('  %-55s %8.3f s' % (k, v))
