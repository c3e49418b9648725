import sys, time, io, contextlib, os, tempfile
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import uberBlast as UB, synth, configure
rng = np.random.default_rng(1)
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
names, seqs = synth.make_genes(ng, 0, seed=11)
d = tempfile.mkdtemp()
os.chdir(d)
with open('genes.fa', 'w') as f:
    for n, s in zip(names, seqs): f.write('>%s\n%s\n' % (n, s.decode()))
contig = []
for k, s in enumerate(seqs[::4]):        # one allele per family on the genome
    contig.append(bytes(rng.choice(list(b'ACGT'), int(rng.integers(50, 300))).tolist()))
    contig.append(s if k % 2 else configure.rc(s.decode()).encode())
genome = b''.join(contig)
with open('genome.fa', 'w') as f: f.write('>1:ctg\n%s\n' % genome.decode())
print('genome nt', len(genome), 'genes', ng)
argv = '-r genome.fa -q genes.fa -f -m -O --diamond --min_id 0.4 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11'.split()
for rep in range(2):
    t0 = time.perf_counter()
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    with contextlib.redirect_stderr(io.StringIO()):
        tab, ovl = UB.uberBlast(argv)
    pr.disable()
    print('rep', rep, 'rows', tab.shape, 'overlaps', ovl.shape, 'seconds %.3f' % (time.perf_counter() - t0))
ps = pstats.Stats(pr); ps.sort_stats('cumulative').print_stats(14)
