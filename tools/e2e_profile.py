"""cProfile of the reference's own hot call through the drop-in (PEPPAN.py:229-230: uberBlast --blastn --diamond -s 1 -e 3,3 on the
exemplar FASTA against itself) on the bench workload - where the host time of `uberblast_e2e_ms` goes.  python3 tools/e2e_profile.py [n_genes]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    from peppan_amd import synth, uberBlast as UB
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    names, seqs = synth.make_genes(n, 1002, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, 'exemplar.fa')
        with open(fa, 'w') as f:
            for i in order:
                f.write('>%s\n%s\n' % (names[i], seqs[i].decode()))
        argv = ('-r %s -q %s --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 1 --min_ratio 0.25 -e 3,3 -p --gtable 11' % (fa, fa)).split()
        with contextlib.redirect_stderr(io.StringIO()):
            UB.uberBlast(argv)
            UB.uberBlast(argv)
            t = time.perf_counter()
            tab = UB.uberBlast(argv)
            dt = time.perf_counter() - t
            pr = cProfile.Profile()
            pr.enable()
            UB.uberBlast(argv)
            pr.disable()
    print('uberBlast: %.1f ms, %d rows' % (dt * 1e3, tab.shape[0]))
    for key in ('cumulative', 'tottime'):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        print(s.getvalue())


if __name__ == '__main__':
    main()
