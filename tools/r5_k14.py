"""round 5: K14 (pair_support) and the uploads in isolation"""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import synth, uberBlast as UB, _native as N, pipeline as PL
names, seqs = synth.make_genes(10000, 1002, seed=355)
packed = N._pack([s for s in seqs])
ctx = UB.get_context()
for rep in range(4):
    t = time.perf_counter(); ctx.set_query_nt(packed, 11); t1 = time.perf_counter(); ctx.set_ref_nt(packed, 6, 11); t2 = time.perf_counter()
    print('set_query_nt %.2f ms, set_ref_nt %.2f ms' % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
prio = {i: [0, -len(s), i] for i, s in enumerate(seqs)}
params = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=2, match_frag_prop=0.25, gtable=11, clust_identity=0.9, clust_match_prop=0.8,
              incompleteCDS='', match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
orig = ctx.pair_support
calls = []
def wrapped(*a):
    calls.append(a)
    t = time.perf_counter(); r = orig(*a); print('pair_support inside get_similar_pairs: %.2f ms' % ((time.perf_counter() - t) * 1e3))
    return r
ctx.pair_support = wrapped
with tempfile.TemporaryDirectory() as tmp:
    ex = os.path.join(tmp, 'ex.clust.exemplar')
    np.save(os.path.join(tmp, 'ex.clust.npy'), np.zeros((0, 3), dtype=int))
    for rep in range(2):
        with open(ex, 'w') as f:
            for i, s in enumerate(seqs):
                f.write('>%d\n%s\n' % (i, s.decode()))
        with contextlib.redirect_stderr(io.StringIO()):
            PL.get_similar_pairs(ex, prio, dict(params, clust=ex))
a = calls[-1]
print('rows', len(a[0]), 'arena', len(a[1]), 'groups', len(a[2]) - 1)
for rep in range(4):
    t = time.perf_counter(); orig(*a); print('pair_support again, same inputs: %.2f ms' % ((time.perf_counter() - t) * 1e3))
for gap in (0.002, 0.005, 0.01, 0.02, 0.05):
    ts = []
    for rep in range(4):
        time.sleep(gap)
        t = time.perf_counter(); orig(*a); ts.append((time.perf_counter() - t) * 1e3)
    print('pair_support after %.0f ms of idle GPU: %s ms' % (gap * 1e3, ' '.join('%.2f' % x for x in ts)))
