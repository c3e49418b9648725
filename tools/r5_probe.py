"""round 5: where does the main stream's pending work inside get_similar_pairs come from?  a tiny synchronising call (K11 over two intervals) as a probe"""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import synth, uberBlast as UB, _native as N, pipeline as PL
names, seqs = synth.make_genes(10000, 1002, seed=355)
ctx = UB.get_context()


def probe(tag):
    t = time.perf_counter()
    if os.environ.get('PROBE') == 'components':
        ctx.components(4, np.array([0, 1], dtype=np.uint32), np.array([1, 2], dtype=np.uint32))
    else:
        ctx.overlaps(np.zeros(2, np.int32), np.array([1, 5]), np.array([10, 20]), np.array([0, 1]), 300., 0.6)
    print('   probe %-28s %.2f ms' % (tag, (time.perf_counter() - t) * 1e3))


prio = {i: [0, -len(s), i] for i, s in enumerate(seqs)}
params = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=2, match_frag_prop=0.25, gtable=11, clust_identity=0.9, clust_match_prop=0.8,
              incompleteCDS='', match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
for name in ('_self_search', '_classify_rows'):
    f = getattr(PL, name)
    def g(*a, _f=f, _n=name, **k):
        r = _f(*a, **k); probe('after ' + _n); probe('again'); return r
    setattr(PL, name, g)
for cls, name in ((UB.RunBlast, '_post'), (UB.RunBlast, '_run_tools'), (UB.RunBlast, '_rescore_table')):
    f = getattr(cls, name)
    def g(self, *a, _f=f, _n=name, **k):
        r = _f(self, *a, **k); probe('end of ' + _n); return r
    setattr(cls, name, g)
from peppan_amd.hittable import HitTable
for name in ('fix_end', 'final_order', 'take'):
    f = getattr(HitTable, name)
    def g(self, *a, _f=f, _n=name, **k):
        r = _f(self, *a, **k); probe('end of HitTable.' + _n); return r
    setattr(HitTable, name, g)
fu = PL.uberBlast
def ub(*a, **k):
    r = fu(*a, **k); probe('end of uberBlast()'); return r
PL.uberBlast = ub
f0 = N.similar_scan
def scan(*a, **k):
    r = f0(*a, **k); probe('after similar_scan'); return r
N.similar_scan = scan
with tempfile.TemporaryDirectory() as tmp:
    ex = os.path.join(tmp, 'ex.clust.exemplar')
    np.save(os.path.join(tmp, 'ex.clust.npy'), np.zeros((0, 3), dtype=int))
    for rep in range(3):
        with open(ex, 'w') as f:
            for i, s in enumerate(seqs):
                f.write('>%d\n%s\n' % (i, s.decode()))
        tm = {}
        with contextlib.redirect_stderr(io.StringIO()):
            PL.get_similar_pairs(ex, prio, dict(params, clust=ex), timing=tm)
        print('call %d: search %.1f decide %.1f' % (rep, tm['search_ms'], tm['decide_ms']), {k: round(v, 2) for k, v in tm['decide_parts_ms'].items()})
