"""round 5: the reference's hot call through the drop-in, timed part by part (no profiler): python tools/r5_e2e.py [n_genes] [reps]"""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import synth, uberBlast as UB, _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
if len(sys.argv) > 3:
    sys.setswitchinterval(float(sys.argv[3]))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
names, seqs = synth.make_genes(n, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
spent = {}


def timed(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        spent[label or name] = spent.get(label or name, 0.) + time.perf_counter() - t
        return r
    setattr(obj, name, g)


with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, 'exemplar.fa')
    with open(fa, 'w') as f:
        for i in order:
            f.write('>%s\n%s\n' % (names[i], seqs[i].decode()))
    argv = ('-r %s -q %s --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 1 --min_ratio 0.25 -e 3,3 -p --gtable 11' % (fa, fa)).split()
    with contextlib.redirect_stderr(io.StringIO()):
        UB.uberBlast(argv); UB.uberBlast(argv)
        ctx = UB.get_context()
        for nm in ('search', 'use_nt_as_residues', 'translate', 'rescore_nt', 'target_meta', 'query_meta'):
            timed(ctx, nm, 'ctx.' + nm)
        for nm in ('table_from_hits', 'cols_gather', 'cols_order', 'cols_fix_end'):
            timed(N, nm)
        from peppan_amd.hittable import HitTable
        for nm in ('to_rows', 'take', 'fix_end', 'final_order'):
            timed(HitTable, nm, 'HitTable.' + nm)
        timed(UB.RunBlast, '_rescore_table'); timed(UB.RunBlast, '_runBlast_table'); timed(UB.RunBlast, '_runDiamond_table'); timed(UB.RunBlast, '_post'); timed(UB.RunBlast, '_run_tools'); timed(UB.RunBlast, '_ensure_nt'); timed(UB.RunBlast, '_load'); timed(UB.RunBlast, 'run')
        t = time.perf_counter()
        for _ in range(reps):
            tab = UB.uberBlast(argv)
        dt = (time.perf_counter() - t) / reps * 1e3
print('uberBlast: %.2f ms per call, %d rows' % (dt, len(tab)))
for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
    print('  %-28s %7.2f ms' % (k, v / reps * 1e3))
