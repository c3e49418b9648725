"""round 5: the search in front of get_similar_pairs - the exemplar file is NEW for every call (the step before rewrote it), so nothing is cached: file read,
sides prepared, both sets uploaded to both contexts, K1, both tools.  Part by part: python tools/r5_cold.py [n_genes] [reps]"""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import synth, uberBlast as UB, _native as N, configure as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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
    text = ''.join('>%s\n%s\n' % (names[i], seqs[i].decode()) for i in order)

    def fresh():
        with open(fa, 'w') as f:
            f.write(text)
        os.utime(fa, ns=(time.time_ns(), time.time_ns()))
    argv = ('-r %s -q %s --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 1 --min_ratio 0.25 -e 3,3 -p --gtable 11' % (fa, fa)).split()
    with contextlib.redirect_stderr(io.StringIO()):
        fresh(); UB.uberBlast(argv, as_table=True); fresh(); UB.uberBlast(argv, as_table=True)
        for c, tag in ((UB.get_context(), 'ctx.'), (UB.get_nucl_context(), 'nctx.')):
            for nm in ('search', 'use_nt_as_residues', 'translate', 'rescore_nt', 'set_query_nt', 'set_ref_nt', 'set_target_groups'):
                timed(c, nm, tag + nm)
        for nm in ('table_from_hits', 'cols_gather', 'cols_order', 'cols_fix_end', '_pack'):
            timed(N, nm)
        timed(UB, 'readFastq'); timed(UB, '_prepare_side'); timed(UB, '_read_cached')
        for nm in ('_rescore_table', '_runBlast_table', '_runDiamond_table', '_post', '_run_tools', '_ensure_nt', '_load', 'run'):
            timed(UB.RunBlast, nm)
        dt = 0.
        for _ in range(reps):
            fresh()
            t = time.perf_counter()
            tab = UB.uberBlast(argv, as_table=True)
            dt += time.perf_counter() - t
        dt = dt / reps * 1e3
print('uberBlast on a new file: %.2f ms per call, %d rows' % (dt, len(tab)))
for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
    print('  %-28s %7.2f ms' % (k, v / reps * 1e3))
