import sys, os, io, contextlib, tempfile, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import synth, uberBlast as UB
n_genes, n_genomes = int(sys.argv[1]), int(sys.argv[2])
os.chdir(tempfile.mkdtemp())
names, seqs = synth.make_genes(n_genes, 0, seed=355)
with open('ex.fa', 'w') as f:
    for i, s in enumerate(seqs[::2]): f.write('>%d\n%s\n' % (i, s.decode()))
files = []
for name, contig, ann in synth.make_genomes(seqs, n_genomes):
    open(name + '.fa', 'w').write('>%s:c1\n%s\n' % (name, contig.decode())); files.append(name + '.fa')
flags = '-q ex.fa -f -m -O --blastn --diamond --min_id 0.4 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11'
with contextlib.redirect_stderr(io.StringIO()):
    UB.uberBlastBatch(files[:2], flags.split())
    pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
    res = UB.uberBlastBatch(files, flags.split())
    dt = time.perf_counter() - t0; pr.disable()
print('%d genomes %.2f s (%.3f s/genome), rows/genome %.0f' % (n_genomes, dt, dt / n_genomes, np.mean([r[0].shape[0] for r in res])))
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
