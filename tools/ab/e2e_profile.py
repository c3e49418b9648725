"""cProfile of the reference's hot call through the drop-in: uberBlast('-r CL -q CL --blastn --diamond -s 1 ... -e 3,3') on the 10 000-gene set"""
import sys, os, tempfile, io, contextlib, cProfile, pstats, time
sys.path.insert(0, '.')
from peppan_amd import synth, uberBlast as UB
names, seqs = synth.make_genes(10000, 1002, seed=355)
d = tempfile.mkdtemp(); fa = os.path.join(d, 'CL.fa')
with open(fa, 'w') as f:
    for n, s in zip(names, seqs): f.write('>%s\n%s\n' % (n, s.decode()))
argv = ('-r %s -q %s --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 1 --min_ratio 0.25 -e 3,3 -p --gtable 11' % (fa, fa)).split()
with contextlib.redirect_stderr(io.StringIO()):
    UB.uberBlast(argv); UB.uberBlast(argv)
    t = time.perf_counter()
    for _ in range(5): tab = UB.uberBlast(argv)
    print('%.1f ms per call, %d rows' % ((time.perf_counter() - t) / 5 * 1e3, len(tab)))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): UB.uberBlast(argv)
    pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
